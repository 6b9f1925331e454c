"""Thin re-export so tools/ scripts can run without installing the package."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unast_amd.portable import portable_state_dict, synth_batch, portable_tensor  # noqa: F401,E402
