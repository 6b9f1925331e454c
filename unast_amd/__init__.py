"""unast_amd: MI355X-native implementation of UNAST's adversarial speech/text train-step hot path.

The HIP library (csrc/ -> libunast_hip.so) is loaded lazily by unast_amd._lib on first kernel use;
importing the package itself needs no GPU.
"""
__version__ = "0.1.0"
