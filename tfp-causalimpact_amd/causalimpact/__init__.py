"""MI355X-native drop-in for the hot path of google/tfp-causalimpact.

Same public names as /root/reference/causalimpact/__init__.py:29-37.  `plot` is out of
scope; `summary` renders the text report.
"""
__version__ = "0.2.0+mi355x.r1"

from causalimpact.causalimpact_lib import CausalImpactAnalysis
from causalimpact.causalimpact_lib import CausalImpactPosteriorSamples
from causalimpact.causalimpact_lib import DataOptions
from causalimpact.causalimpact_lib import fit_causalimpact
from causalimpact.causalimpact_lib import InferenceOptions
from causalimpact.causalimpact_lib import ModelOptions
from causalimpact.causalimpact_lib import Seasons
from causalimpact.indices import InputDateType
from causalimpact.summary import summary
from causalimpact.summary import summary_numbers
from causalimpact.batch import CausalImpactBatchAnalysis
from causalimpact.batch import fit_causalimpact_batch


def plot(*args, **kwargs):
  """Not part of this build: the reference draws Altair charts
  from `CausalImpactAnalysis.series`, which this build returns with the same schema -- the
  reference's `causalimpact.plot` works on it unchanged."""
  raise NotImplementedError(
      "plot() is not provided by the MI355X build; pass the analysis to the reference's "
      "causalimpact.plot (same CausalImpactAnalysis.series schema)")
