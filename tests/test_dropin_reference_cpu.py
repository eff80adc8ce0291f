"""Drop-in boundary: the reference's own `seva/eval.py` imported ON TOP of this package.

With `SEVA_REFERENCE_PATH` set, `seva/__init__.py` and `seva/modules/__init__.py` append the reference's directories to
their `__path__` (own modules keep precedence) and `seva.geometry` re-exports the reference's helpers around the HIP
`get_plucker_coordinates`.  The reference's callers then bind to the product's operator API:

  * `seva.eval.GradioTrackedSampler` (reference `seva/eval.py:1037-1089`) subclasses the PRODUCT's `EulerEDMSampler`;
  * `seva.eval.create_samplers` (reference `seva/eval.py:1092-1149`) builds product guiders and samplers;
  * `demo.py:29-52`'s imports (`seva.model`, `seva.sampling`, `seva.modules.autoencoder`, `seva.utils`) resolve to the product,
    `seva.eval` / `seva.data_io` to the reference.

Build-container only (the reference does not travel to the GPU box); runs in a child process so that the shadowed package
does not leak into the rest of the suite.  The UI / IO packages the reference imports but these code paths never touch
(gradio, colorama, imageio, torchvision, roma, cv2) get inert stand-ins when they are not installed.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stable-virtual-camera_amd")
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "seva", "eval.py")),
                                reason="the reference checkout is only present in the build container")

CHILD = textwrap.dedent(
    r"""
    import json, os, sys, threading, types

    class _Inert(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return _Inert(self.__name__ + "." + name)
        def __call__(self, *a, **k):
            return _Inert("call")
        def __or__(self, o):
            return self
        def __ror__(self, o):
            return self

    for m in ["roma", "gradio", "colorama", "imageio", "imageio.v3", "torchvision", "torchvision.transforms",
              "torchvision.transforms.functional", "cv2"]:
        try:
            __import__(m)
        except Exception:
            sys.modules.setdefault(m, _Inert(m))

    import torch
    import seva, seva.modules
    import seva.sampling as S
    import seva.model as Mdl
    import seva.utils as U
    import seva.geometry as G
    import seva.modules.autoencoder as AE
    import seva.eval as E          # the REFERENCE's file, through the extended __path__

    pkg, ref = os.environ["PKG"], os.environ["SEVA_REFERENCE_PATH"]
    out = {}
    out["files"] = {"sampling": S.__file__, "model": Mdl.__file__, "utils": U.__file__, "geometry": G.__file__,
                    "autoencoder": AE.__file__, "eval": E.__file__}
    out["seva_path"] = list(seva.__path__)
    out["modules_path"] = list(seva.modules.__path__)
    # eval.py's own imports bound to the product's classes
    out["eval_binds_product"] = (E.EulerEDMSampler is S.EulerEDMSampler and E.MultiviewCFG is S.MultiviewCFG
                                 and E.MultiviewTemporalCFG is S.MultiviewTemporalCFG and E.VanillaCFG is S.VanillaCFG
                                 and E.get_plucker_coordinates is G.get_plucker_coordinates and E.seed_everything is U.seed_everything)
    out["tracked_mro1_is_product"] = E.GradioTrackedSampler.__mro__[1] is S.EulerEDMSampler
    disc = S.DDPMDiscretization()
    plain = E.create_samplers([0, 1, 2], disc, [21, 21, 21], 7, cfg_min=1.2, device="cpu")
    out["plain"] = [[type(s).__module__, type(s).__name__, type(s.guider).__module__, type(s.guider).__name__, s.num_steps] for s in plain]
    tracked = E.create_samplers(1, disc, None, 5, cfg_min=1.2, device="cpu", abort_event=threading.Event())
    out["tracked"] = [[type(s).__module__, type(s).__name__, isinstance(s, S.EulerEDMSampler), type(s.guider).__module__,
                       hasattr(s, "abort_event")] for s in tracked]
    # the subclass protocol GradioTrackedSampler drives (eval.py:1053-1089).  Host-only parts run here; the part that touches
    # latents must FAIL LOUDLY without a GPU (no CPU fallback behind the drop-in boundary)
    try:
        tracked[0].prepare_sampling_loop(torch.ones(3, 4, 8, 8), {"a": 1}, {"b": 2}, 5)
        out["cpu_latents"] = "no error"
    except Exception as e:
        out["cpu_latents"] = type(e).__name__
    sig = disc(5, device="cpu")
    out["loop"] = [int(sig.shape[0]), bool(float(sig[0]) > float(sig[1]) > 0.0), float(sig[-1]), len(list(tracked[0].get_sigma_gen(6, False))),
                   [tracked[0].s_churn, tracked[0].s_tmin, tracked[0].s_tmax]]
    out["plucker_module"] = G.get_plucker_coordinates.__module__
    out["reexported"] = {n: getattr(getattr(G, n, None), "__module__", None) for n in
                         ("get_preset_pose_fov", "generate_spiral_path", "generate_interpolated_path", "get_lookat", "normalize_scene")}
    out["own_helpers"] = {n: getattr(G, n).__module__ for n in ("to_hom_pose", "get_default_intrinsics")}
    # get_camera_dist: eval.py imports it from seva.geometry (the reference's, re-exported); the product's guiders use their own
    out["camera_dist"] = [G.get_camera_dist.__module__, S.get_camera_dist.__module__,
                          bool(torch.equal(G.get_camera_dist(torch.eye(4)[None], torch.eye(4)[None].repeat(2, 1, 1)),
                                           S.get_camera_dist(torch.eye(4)[None], torch.eye(4)[None].repeat(2, 1, 1))))]
    # planner functions of the reference run against the product package (pure host logic)
    out["pad_indices"] = [list(map(int, v)) for v in E.pad_indices([0, 5], [1, 2, 3], T=8, padding_mode="last")]
    print("RESULT " + json.dumps(out))
    """
)


def _run_child():
    env = dict(os.environ)
    env["SEVA_REFERENCE_PATH"] = REF
    env["PKG"] = PKG
    env["PYTHONPATH"] = PKG + os.pathsep + REF  # this package BEFORE the reference checkout
    env["PYTHONDONTWRITEBYTECODE"] = "1"        # never write into the (read-only) reference tree
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.fixture(scope="module")
def dropin():
    return _run_child()


def test_hot_path_modules_resolve_to_the_product_and_callers_to_the_reference(dropin):
    f = dropin["files"]
    for name in ("sampling", "model", "utils", "geometry", "autoencoder"):
        assert f[name].startswith(PKG), (name, f[name])
    assert f["eval"].startswith(REF)
    assert dropin["seva_path"][0].startswith(PKG) and os.path.join(REF, "seva") in dropin["seva_path"]
    assert dropin["modules_path"][0].startswith(PKG) and os.path.join(REF, "seva", "modules") in dropin["modules_path"]


def test_reference_eval_binds_the_product_operator_api(dropin):
    assert dropin["eval_binds_product"]
    assert dropin["tracked_mro1_is_product"]  # GradioTrackedSampler.__mro__[1] is seva.sampling.EulerEDMSampler (product)


def test_create_samplers_builds_product_guiders_and_samplers(dropin):
    assert dropin["plain"] == [["seva.sampling", "EulerEDMSampler", "seva.sampling", "VanillaCFG", 7],
                               ["seva.sampling", "EulerEDMSampler", "seva.sampling", "MultiviewCFG", 7],
                               ["seva.sampling", "EulerEDMSampler", "seva.sampling", "MultiviewTemporalCFG", 7]]
    assert dropin["tracked"] == [["seva.eval", "GradioTrackedSampler", True, "seva.sampling", True]]
    assert dropin["cpu_latents"] == "SevaNativeError"
    n_sig, decreasing, last, n_iter, churn = dropin["loop"]
    assert n_sig == 6 and decreasing and last == 0.0 and n_iter == 5 and churn == [0.0, 0.0, 999.0]


def test_geometry_is_the_hip_kernel_with_the_reference_presets_reexported(dropin):
    assert dropin["plucker_module"] == "seva.geometry"
    assert dropin["own_helpers"] == {"to_hom_pose": "seva.geometry", "get_default_intrinsics": "seva.geometry"}
    assert dropin["camera_dist"] == ["seva._reference_geometry", "seva.sampling", True]
    for name, mod in dropin["reexported"].items():
        assert mod == "seva._reference_geometry", (name, mod)


def test_reference_planner_runs_on_top_of_the_product_package(dropin):
    # reference seva/eval.py:44-82 with padding_mode="last": both index lists padded to T by repeating the last entry
    inp, tst, imap, tmap = dropin["pad_indices"]
    assert inp == [0, 4, 5, 6, 7] and tst == [1, 2, 3]
    assert imap == [0, -1, -1, -1, 1, 1, 1, 1] and tmap == [-1, 0, 1, 2, -1, -1, -1, -1]
