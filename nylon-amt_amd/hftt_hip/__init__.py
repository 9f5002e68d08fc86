"""MI355X-native hFT-Transformer path: ctypes binding of libhftt_hip.so + host engine.

Import layout mirrors the reference's ``hftt_code/`` directory: put ``nylon-amt_amd/`` on ``sys.path`` and import
``model.model_spec2midi`` / ``model.amt`` / ``training.train`` exactly as the reference's scripts do.
"""
from ._capi import HfttError, lib, LIB_PATH   # noqa: F401
