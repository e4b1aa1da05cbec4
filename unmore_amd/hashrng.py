"""Platform-independent counter-based RNG (FNV-1a key + splitmix64 mix).

Used for synthetic weights/inputs so that golden fixtures made in the build
container (where the reference's Python can run) can be regenerated bit-exactly
on the GPU box from a name and a shape alone -- `torch.manual_seed` streams are
not guaranteed stable across builds/devices.  Pure numpy integer arithmetic.
"""
import numpy as np

_M64 = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(name: str, shape) -> np.ndarray:
    """float32 array in [0,1) determined only by (name, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(fnv1a64(name))
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + key
    z = _splitmix64(ctr)
    u = (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return u.reshape(shape)


def uniform(name: str, shape, lo: float, hi: float) -> np.ndarray:
    return (uniform01(name, shape) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)


def hash_init(name: str, shape, tag: str = "w0") -> np.ndarray:
    """Deterministic synthetic value for a state-dict entry (variance-preserving
    uniform weights, small biases, LayerNorm scales near 1)."""
    shape = tuple(int(s) for s in shape)
    key = tag + ":" + name
    if name.endswith(".bias"):
        return uniform(key, shape, -0.05, 0.05)
    if name.endswith("cls_token") or name.endswith("pos_embed"):
        return uniform(key, shape, -0.3, 0.3)
    if len(shape) == 1:  # LayerNorm scale
        return uniform(key, shape, 0.9, 1.1)
    if ("act_postprocess1.4." in name) or ("act_postprocess2.4." in name):
        fan_in = shape[0]  # ConvTranspose2d [in,out,k,k] with stride == kernel
    else:
        fan_in = int(np.prod(shape[1:]))
    a = float(np.sqrt(3.0 / fan_in))
    if shape[0] <= 2:  # final head layers: keep tanh out of saturation
        a *= 0.25 if shape[0] == 2 else 0.05
    return uniform(key, shape, -a, a)
