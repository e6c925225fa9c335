"""MI355X-native IndexTTS2-style semantic-token LM hot path (prefill + paged decode).

The directory name carries a hyphen (it is the reference repo's name + `_amd`), so import it with
`importlib.import_module("maxtext-indextts2_amd")` or through the root-level alias module
`maxtext_indextts2_amd.py`.
"""
__version__ = "0.1.0"
