"""Drop-in for the reference's LBDRNloss module."""
from lbdrn_hip.model import LBDRNLoss  # noqa: F401
