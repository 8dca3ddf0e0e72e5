"""Feature switches shared by encoder and decoder (they are NOT stored in the bitstream, so both
sides must run with the same values; same names and defaults as ref constants.py:3-14)."""

USE_COORDINATES = False   # prepend normalised (row, col) coordinates
EMBEDDING = False         # ... expanded by sin/cos of SIGMA**j * pi * coord, j < N_FREQ
SIGMA = 1.4
N_FREQ = 12
USE_COLORS = True         # (2D+1)^2 neighbourhood of normalised high bits per band
RELATIVE = True           # subtract the centre pixel from its neighbourhood
# Not in the reference's constants.py: the reference switches its hidden activation by editing a source line
# (`activation=torch.nn.ReLU() # Default: Sine`, commented out at ref encode.py:75 and decode.py:108); this name stands for
# that edit.  "sine" = Sine(w0 = 30), the fused MFMA kernels; "relu" = torch.nn.ReLU(): the same fused kernels since round 6 (the activation is their template argument); a .bin written with it says so in its header.
HIDDEN_ACTIVATION = "sine"
