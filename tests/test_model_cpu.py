"""CPU checks of the host side: state-dict schema parity with the reference, the
C-ABI library exports, and loud failure without a GPU."""
import ctypes
import os
import re
from argparse import Namespace

import pytest
import torch


def _schema(path):
    out = []
    for line in open(path):
        parts = line.split()
        out.append((parts[0], tuple(int(x) for x in parts[1:])))
    return out


@pytest.mark.parametrize("backbone,fname", [("dpt_large", "schema_dpt_large.txt"), ("dpt_base", "schema_dpt_base.txt"),
                                            ("dpt_tiny", "schema_dpt_tiny.txt")])
def test_state_dict_schema_matches_reference(golden_dir, backbone, fname):
    from unmore_amd.objectness_net import ObjectnessNet
    args = Namespace(use_bg_sdf=True, sdf_activation="tanh")
    with torch.device("meta"):
        net = ObjectnessNet("cpu", 128, backbone, args)
    mine = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert mine == _schema(os.path.join(golden_dir, fname))


def test_head_variants_and_errors():
    from unmore_amd.objectness_net import ObjectnessNet
    with torch.device("meta"):
        n = ObjectnessNet("cpu", 128, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="relu"))
        assert "sdf_prediction_head.6.weight" in n.state_dict()
        n = ObjectnessNet("cpu", 128, "dpt_tiny", Namespace(use_bg_sdf=False, sdf_activation="tanh"))
        assert "sdf_prediction_head.6.weight" in n.state_dict()
        n = ObjectnessNet("cpu", 128, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation=None))
        assert "sdf_prediction_head.3.weight" in n.state_dict()
        with pytest.raises(NotImplementedError):
            ObjectnessNet("cpu", 128, "no_such_backbone", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
        with pytest.raises(NotImplementedError):
            ObjectnessNet("cpu", 128, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="bogus"))


def test_nograd_names_match_reference(golden_dir):
    from unmore_amd.objectness_net import ObjectnessNet
    with torch.device("meta"):
        net = ObjectnessNet("cpu", 128, "dpt_large", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    ref = sorted(l.strip() for l in open(os.path.join(golden_dir, "nograd_dpt_large.txt")) if l.strip())
    assert sorted(net.nograd_names()) == ref


def test_cpu_tensors_fail_loudly():
    from unmore_amd.objectness_net import ObjectnessNet
    net = ObjectnessNet("cpu", 64, "dpt_tiny", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(images=torch.zeros(1, 3, 64, 64))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports every function include/umr.h declares."""
    from unmore_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "umr.h")).read()
    declared = set(re.findall(r"\b(umr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/umr.h but not exported"
    assert declared == set(_lib.exported_symbols())
    assert lib.umr_version() >= 100


def test_argument_errors_are_reported_without_a_gpu():
    """Argument checks run before any HIP call: a null descriptor and an undersized split-K workspace come back as
    UMR_ERR_INVALID with a message (no compute is attempted on this box)."""
    from unmore_amd import _lib
    lib = _lib.lib()
    assert lib.umr_gemm_nt(None, None) != 0
    assert b"null descriptor" in lib.umr_last_error_string()
    need = lib.umr_gemm_nt_workspace()
    assert need >= 16384 + 32 * 1024 * 1024
    d = _lib.GemmDesc()
    buf = ctypes.create_string_buffer(64)
    assert lib.umr_gemm_nt_ws(ctypes.byref(d), ctypes.cast(buf, ctypes.c_void_p), 64, None) != 0
    assert b"workspace" in lib.umr_last_error_string()


def test_debug_options_are_read_at_load_and_changed_only_through_the_entry_point():
    """No entry point reads the environment on its launch path (round-5 review): the UMR_* A/B options are read once, when the
    library is loaded, and changed through umr_set_debug_option.  A child process checks both halves: an option present in the
    environment at load is seen; changing the environment afterwards changes nothing; the setter does; unknown names are refused."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import os, sys
sys.path.insert(0, %r)
from unmore_amd import _lib, ops
import ctypes
lib = _lib.lib()
def get(name):
    v, s = ctypes.c_int(0), ctypes.c_int(0)
    assert lib.umr_get_debug_option(name.encode(), ctypes.byref(v), ctypes.byref(s)) == 0
    return v.value if s.value else None
assert get("UMR_NT_SPLITK") == 3 and get("UMR_NT_ORDER") == ord("m") and get("UMR_GEMM_TILE") is None
os.environ["UMR_NT_SPLITK"] = "7"
os.environ["UMR_GEMM_TILE"] = "128"
assert get("UMR_NT_SPLITK") == 3 and get("UMR_GEMM_TILE") is None          # the environment is not consulted again
assert ops.set_debug_option("UMR_GEMM_TILE", 256) is None and get("UMR_GEMM_TILE") == 256
assert ops.set_debug_option("UMR_GEMM_TILE", None) == 256 and get("UMR_GEMM_TILE") is None
assert ops.set_debug_option("UMR_NT_ORDER", "n") == ord("m") and get("UMR_NT_ORDER") == ord("n")
assert lib.umr_set_debug_option(b"UMR_NO_SUCH_OPTION", b"1") != 0 and b"unknown option" in lib.umr_last_error_string()
assert lib.umr_set_debug_option(None, b"1") != 0
print("ok")
""" % root
    env = dict(os.environ, UMR_NT_SPLITK="3", UMR_NT_ORDER="m")
    env.pop("UMR_GEMM_TILE", None)
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
    # and the sources themselves: getenv only in once-per-process initialisers (static locals / first-use atomics), never in a launch path
    csrc = os.path.join(root, "unmore_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for ln in open(os.path.join(csrc, f)):
                if "getenv(" in ln and not ln.lstrip().startswith("//"):
                    assert ("static const" in ln or "static inline int umr_env_int" in ln or f == "umr_api.hip"), (f, ln.strip())


def test_flat_layout_buckets_follow_backward_completion_order():
    """trainer.flat_layout (shared by TrainStep and bench.py --rehearse): every grad-receiving parameter has a 256-byte aligned
    slot, parameters the reference never back-propagates into have none, and the bucket boundaries follow the order in which
    backward completes the stages (heads, refinenets, reassemble, blocks last..first, embeddings)."""
    from argparse import Namespace
    from unmore_amd.objectness_net import ObjectnessNet
    from unmore_amd.trainer import flat_layout
    net = ObjectnessNet("cpu", 64, "dpt_base", Namespace(use_bg_sdf=True, sdf_activation="tanh"))
    offs, bounds, stage_bucket = flat_layout(net)
    named = dict(net.named_parameters())
    assert set(offs) == set(named) - net.nograd_names()
    assert all(o % 64 == 0 for o in offs.values()) and bounds[0] == 0 and bounds == sorted(bounds)
    assert len(bounds) - 1 == 16 and list(stage_bucket)[:3] == ["heads", "refine", "reassemble"] and list(stage_bucket)[-1] == "embed"
    assert list(stage_bucket)[3] == "block11" and stage_bucket["block0"] == 14
    assert 115_000_000 < bounds[-1] < 116_000_000          # 115,400,899 elements + alignment padding
    for n, o in offs.items():                               # every slot lies inside its stage's bucket
        k = [i for i in range(16) if bounds[i] <= o < bounds[i + 1]]
        assert len(k) == 1 and o + named[n].numel() <= bounds[k[0] + 1]


def test_graph_and_stream_policies():
    """graphs.wanted / WgradStream.wanted: 'auto' captures small inference calls, and small train steps where the eager schedule would
    use two streams (as a chain of per-stage graphs); large problems run eagerly; the second stream serves small problems only (the
    384x384 batches fill the chip and keep one stream)."""
    from unmore_amd import engine, graphs
    small, big = 20 * 128 * 128, 64 * 384 * 384
    assert graphs.wanted("auto", small) and not graphs.wanted("auto", big)
    assert graphs.wanted("auto", small, train=True) and not graphs.wanted("auto", small, train=True, two_streams=False)
    assert not graphs.wanted("auto", big, train=True)
    assert graphs.wanted("on", big, train=True) and not graphs.wanted("off", small) and not graphs.wanted("off", small, train=True)
    assert engine.WgradStream.wanted(small) and not engine.WgradStream.wanted(big)
    assert graphs.CAPTURE_TYPES == (graphs.Captured, graphs.StagedCaptured) and graphs.lane() is None and graphs.staged() is None


def test_xt_views_share_one_cell():
    """engine_x3.XT: a value held as f32 and / or bf16 planes; views share the cell, so a conversion made through one view is
    visible through every other (no second split pass)."""
    import torch
    from unmore_amd.engine_x3 import XT
    f = torch.arange(24, dtype=torch.float32).view(2, 3, 4)
    x = XT(f=f)
    v = x.view(6, 4)
    assert v.shape == (6, 4) and x.shape == (2, 3, 4) and v.f.data_ptr() == f.data_ptr() and v.p is None
    planes = torch.zeros((6, 12), dtype=torch.bfloat16)
    v.cell[1] = planes                       # what P() stores after split3 on the GPU
    assert x.p.shape == (2, 3, 12) and x.p.data_ptr() == planes.data_ptr() and x.n == 4
    assert x.mask_source().data_ptr() == f.data_ptr()
    x.drop_f()
    assert v.f is None and v.any().data_ptr() == planes.data_ptr() and v.mask_source().shape == (6, 4)


def test_pack_cache_param_groups_and_capture_purge():
    """engine.PackCache on CPU tensors (no stream, no events): a pack derived from SEVERAL parameters (ParamGroup: the collapsed
    head's weight algebra reads eight tensors) is rebuilt when any one of them changes in place; packs a failed capture only
    recorded are purged together with the validity stamps of captures that read them."""
    import torch
    from unmore_amd import graphs
    from unmore_amd.engine import PackCache, ParamGroup
    a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.full((3,), 2.0))
    cache = PackCache()
    built = []

    def build():
        built.append(1)
        return (a.detach() + b.detach()).clone()
    key = ("head", "collapsed", torch.float32)
    v0 = cache.get(key, ParamGroup([a, b]), build)
    assert torch.equal(v0, torch.full((3,), 3.0)) and len(built) == 1
    assert cache.get(key, ParamGroup([a, b]), build) is v0 and len(built) == 1          # hit: same versions, same storages
    with torch.no_grad():
        b.add_(1.0)                                                                      # in-place change of the SECOND member
    v1 = cache.get(key, ParamGroup([a, b]), build)
    assert len(built) == 2 and torch.equal(v1, torch.full((3,), 4.0))
    assert key in cache._o and key not in cache._c                                      # no replay recipe: dropped after an optimizer step
    gen = cache.generation()
    cache.refresh_done()
    assert key not in cache._o and cache.generation() != gen
    # a capture that failed: its entries carry its store
    store = {}
    graphs._state.update(capturing=True, store=store)
    try:
        cache.get(("x", "lin_t", torch.float32), a, lambda: a.detach().clone())
    finally:
        graphs._state.update(capturing=False, store=None)
    assert ("x", "lin_t", torch.float32) in cache._o and cache._o[("x", "lin_t", torch.float32)][6] is store
    gen = cache.generation()
    assert cache.purge_capture(store) == 1 and ("x", "lin_t", torch.float32) not in cache._o
    assert cache.generation({"hit_o": True}) != (gen[0], gen[1])
    assert cache.purge_capture(store) == 0
