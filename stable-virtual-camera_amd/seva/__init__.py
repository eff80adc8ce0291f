"""MI355X-native `seva` package: the multi-view denoising hot path of Stable Virtual Camera.

Ships `seva.model`, `seva.sampling` and `seva.modules.autoencoder` (the operator API that
`demo.py` / `seva/eval.py` of the reference import).  Everything else of the
reference (`seva.eval`, `seva.geometry`, `seva.data_io`, `seva.gui`, ...) is host-side code that
is out of scope here; to run the reference's `demo.py` unchanged, point `SEVA_REFERENCE_PATH`
at a checkout of the reference and those modules resolve from there while the hot path resolves
to this package.
"""

import os as _os

_ref = _os.environ.get("SEVA_REFERENCE_PATH")
if _ref:
    _cand = _os.path.join(_ref, "seva")
    if _os.path.isdir(_cand) and _cand not in __path__:
        __path__.append(_cand)  # this package's own modules keep precedence
