"""unast_amd: MI355X-native implementation of UNAST's adversarial speech/text train-step hot path.

The HIP library (csrc/ -> libunast_hip.so) is loaded lazily by unast_amd._lib on first kernel use;
importing the package itself needs no GPU.
"""
__version__ = "0.1.0"

import os as _os

# The stream schedule of the train step (engine.side_streams: text | speech | discriminator + weight-gradient companions) was
# tuned with HIP's default of 4 hardware queues per device, and the number matters beyond which streams alias: the same
# streams took 37.6 ms/step with 4 queues and 45-56 ms with 3, 5, 6, 8 or 16.  The HIP runtime reads the
# variable when it initialises, i.e. at the first device call, so setting the default here is early enough; an explicit
# setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
