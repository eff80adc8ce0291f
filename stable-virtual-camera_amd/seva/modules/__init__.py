import os as _os

_ref = _os.environ.get("SEVA_REFERENCE_PATH")
if _ref:
    _cand = _os.path.join(_ref, "seva", "modules")
    if _os.path.isdir(_cand) and _cand not in __path__:
        __path__.append(_cand)
