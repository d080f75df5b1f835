"""MI355X-native drop-in for the hot path of google/tfp-causalimpact.

Mirrors /root/reference/causalimpact/__init__.py:29-37 (same public names).
"""
__version__ = "0.2.0+mi355x.r1"
