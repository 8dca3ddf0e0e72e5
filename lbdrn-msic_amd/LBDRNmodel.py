"""Drop-in for the reference's LBDRNmodel module: same class names and signatures, forward on HIP."""
from lbdrn_hip.model import LBDRNModel, Sine, SirenLayer  # noqa: F401
