"""ctypes binding of libsykepic_hip.so (C-ABI: include/sykepic_hip.h).

There is no CPU fallback: if the library is missing or fails to load, every
use of the product path raises.
"""

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SYKEPIC_HIP_LIB", _HERE / "libsykepic_hip.so"))


class LayerDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("k", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("relu", C.c_int32), ("src", C.c_int32),
        ("dst", C.c_int32), ("res", C.c_int32), ("child", C.c_int32), ("p", C.c_float),
        ("name", C.c_char * 96), ("bn", C.c_char * 96),
    ]


class OptimDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32), ("lr", C.c_float * 3), ("beta1", C.c_float), ("beta2", C.c_float),
        ("eps", C.c_float), ("weight_decay", C.c_float), ("momentum", C.c_float),
        ("grad_scale", C.c_float), ("alpha", C.c_float), ("momentum_decay", C.c_float),
        ("lr_decay", C.c_float), ("initial_accumulator_value", C.c_float),
    ]


class Roi(C.Structure):
    _fields_ = [("offset", C.c_int64), ("width", C.c_int32), ("height", C.c_int32)]


class AugOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("i0", C.c_int32), ("i1", C.c_int32), ("pad_", C.c_int32), ("d", C.c_double * 6)]


AUG_FLIP_H, AUG_FLIP_V, AUG_TRANSLATE, AUG_ZOOM, AUG_ROTATE, AUG_BRIGHT = 1, 2, 3, 4, 5, 6


class LayerTime(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("ms", C.c_float), ("flops", C.c_double),
                ("bytes", C.c_double)]


LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
DTYPE_F32, DTYPE_I64, DTYPE_U8 = 0, 1, 2
OPT_SGD, OPT_ADAM, OPT_ADAMW, OPT_RMSPROP, OPT_ADAGRAD, OPT_ADAMAX, OPT_NADAM, OPT_RADAM, OPT_ADADELTA = range(9)
OPT_ASGD, OPT_RPROP = 9, 10

# every symbol include/sykepic_hip.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "spk_last_error": (C.c_char_p, []),
    "spk_version": (C.c_char_p, []),
    "spk_model_create": (C.c_int, [C.POINTER(LayerDesc), C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(_P)]),
    "spk_model_destroy": (None, [_P]),
    "spk_model_set_stream": (C.c_int, [_P, _P]),
    "spk_model_num_params": (C.c_int, [_P]),
    "spk_model_param_info": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "spk_model_load_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "spk_model_read_param": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "spk_model_set_requires_grad": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "spk_model_set_param_group": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "spk_model_set_infer_dtype": (C.c_int, [_P, C.c_int]),
    "spk_model_set_precision": (C.c_int, [_P, C.c_int, C.c_int]),
    "spk_model_set_split_ops": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "spk_model_act_means_size": (C.c_int64, [_P]),
    "spk_model_calibrate_act_means": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "spk_model_get_act_means": (C.c_int, [_P, _P, C.c_int64]),
    "spk_model_set_act_means": (C.c_int, [_P, _P, C.c_int64]),
    "spk_model_set_zero_sum": (C.c_int, [_P, C.c_int]),
    "spk_op_conv_dgrad_bn_backward": (C.c_int, [_P, _P, _P, C.c_int] + [_P] * 8 + [C.c_int] * 8 + [_P, _P, _P]),
    "spk_op_conv1x1_chain": (C.c_int, [_P] * 10 + [C.c_int] * 8 + [_P]),
    "spk_op_conv1x1_dual": (C.c_int, [_P] * 9 + [C.c_int] * 12 + [_P]),
    "spk_op_zero_sum_round": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.c_int, _P]),
    "spk_model_set_fp8": (C.c_int, [_P, C.c_int]),
    "spk_model_calibrate_fp8": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "spk_model_set_bn": (C.c_int, [_P, C.c_float, C.c_float]),
    "spk_model_set_fp8_blocks": (C.c_int, [_P, _P, C.c_int]),
    "spk_model_num_fp8_blocks": (C.c_int, [_P]),
    "spk_model_set_seed": (C.c_int, [_P, C.c_uint64]),
    "spk_forward_infer": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    _P]),
    "spk_eval_step": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "spk_train_forward_backward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             _P, _P, _P]),
    "spk_optim_step": (C.c_int, [_P, C.POINTER(OptimDesc)]),
    "spk_model_grad_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "spk_model_set_grad_ready_callback": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "spk_model_read_grad": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "spk_model_read_activation": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64]),
    "spk_model_read_activation_grad": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64]),
    "spk_op_conv_bn_train_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 9 + [_P]),
    "spk_op_bn_backward": (C.c_int, [_P] * 10 + [C.c_int] * 4 + [_P]),
    "spk_op_conv_dgrad": (C.c_int, [_P, _P, _P] + [C.c_int] * 9 + [_P]),
    "spk_op_conv_wgrad": (C.c_int, [_P, _P, _P] + [C.c_int] * 8 + [_P]),
    "spk_op_pw_fp8": (C.c_int, [_P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_float, C.c_float, _P]),
    "spk_op_dwconv": (C.c_int, [_P, _P, _P, _P, _P, _P] + [C.c_int] * 8 + [_P]),
    "spk_op_conv1x1": (C.c_int, [_P, _P, _P, _P, _P, _P] + [C.c_int] * 9 + [_P]),
    "spk_op_conv1x1_num_configs": (C.c_int, []),
    "spk_op_conv3x3": (C.c_int, [_P, _P, _P, _P, _P, _P] + [C.c_int] * 8 + [_P]),
    "spk_op_conv3x3_num_configs": (C.c_int, []),
    "spk_op_bottleneck": (C.c_int, [_P] * 11 + [C.c_int] * 6 + [_P, _P, _P] + [_P] * 4 + [C.c_int]),
    "spk_preprocess_rois": (C.c_int, [_P, C.c_int64, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "spk_predict_rows": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_float, _P, _P, _P]),
    "spk_augment_batch": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "spk_model_profile_infer": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(LayerTime), C.c_int]),
    "spk_model_profile_train": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P,
                                          C.c_int, C.POINTER(LayerTime), C.c_int]),
}

GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int64, C.c_int64)

_lib = None


def _default_tune_cache():
    """The kernel tuner's winners persist per machine and per build of the library unless the caller says otherwise:
    `SPK_TUNE_CACHE=<file>` names the file, `SPK_TUNE_CACHE=off` (or 0 / empty) keeps the choices in the process only.
    Default: $XDG_CACHE_HOME (or ~/.cache)/sykepic_hip/tune-<size>-<mtime of the .so>.txt - a rebuilt library starts
    a new file.  Without it every `sykepic prob` process re-times ~40 candidates for each convolution shape before its
    first batch (0.3 s of a 0.6 s single-sample run, tools/e2e_prob.py)."""
    import os
    v = os.environ.get("SPK_TUNE_CACHE")
    if v is not None:
        if v.strip().lower() in ("", "0", "off", "none"):
            del os.environ["SPK_TUNE_CACHE"]
        return
    if not os.path.exists("/dev/kfd"):
        return         # no AMD GPU on this host (build / CPU-test containers): nothing will be tuned, touch nothing
    try:
        st = LIB_PATH.stat()
        base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
        d = os.path.join(base, "sykepic_hip")
        os.makedirs(d, exist_ok=True)
        if os.access(d, os.W_OK):
            path = os.path.join(d, f"tune-{st.st_size:x}-{int(st.st_mtime):x}.txt")
            os.environ["SPK_TUNE_CACHE"] = path
            seed_tune_cache(path)
    except OSError:
        pass   # no writable cache directory: tune per process


TUNE_SEED = Path(__file__).resolve().parent / "tune_seed_gfx950.txt"


def seed_tune_cache(path, seed=None):
    """A tuning cache that does not exist yet starts as a copy of the winners measured on one MI355X for the shapes of
    the shipped benchmarks (`tune_seed_gfx950.txt`: comment lines are skipped by the loaders): the first `sykepic prob`
    of a fresh installation then times nothing for those shapes; every other problem is tuned on first use and appended,
    as before.  Every candidate of a problem computes the same values, so a seed can only cost speed on a machine that
    would have chosen differently.  `SPK_TUNE_SEED=0` starts from an empty cache.  Returns True when it wrote the file."""
    import os
    seed = Path(seed) if seed is not None else TUNE_SEED
    if os.environ.get("SPK_TUNE_SEED", "1").strip().lower() in ("0", "off", "no") or os.path.exists(path) or not seed.is_file():
        return False
    tmp = f"{path}.{os.getpid()}.tmp"
    try:
        with open(tmp, "w") as f:
            f.write(seed.read_text())
        os.replace(tmp, path)      # atomic: ranks of one job race for the same file, any winner is complete
        return True
    except OSError:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        return False


def load():
    """Load the HIP library (once). Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.is_file():
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP library is not built "
            "(run syke-pic_amd/csrc/build.sh or __graft_entry__.build()); "
            "there is no CPU fallback on the product path")
    _default_tune_cache()
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().spk_last_error().decode(errors="replace")
        raise RuntimeError(f"libsykepic_hip error {rc}: {msg}")
