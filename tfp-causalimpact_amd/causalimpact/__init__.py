"""MI355X-native drop-in for the hot path of google/tfp-causalimpact.

Same public names as /root/reference/causalimpact/__init__.py:29-37: `fit_causalimpact`, the
option / result types, `summary` (text report) and `plot` (Vega-Lite or matplotlib).
"""
__version__ = "0.2.0+mi355x.r2"

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


from causalimpact.plot import plot
