"""ctypes binding of the C-ABI library (include/gpt_hip.h).  There is no CPU fallback: if the HIP
library is missing, or a tensor is not on a GPU, the call fails loudly."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPT_HIP_LIB") or os.path.join(_HERE, "lib", "libgpt_hip.so")   # override: kernel experiments

F32, BF16 = 0, 1
DP_I64, DP_ID_BYTES = 2, 128
EPI_STORE, EPI_GELU, EPI_RESADD, EPI_STORE_F32, EPI_GELU_AUX, EPI_DGELU, EPI_ACCUM_F32 = range(7)
MAX_TOKENS = 40
SRC_X, SRC_XPREV, SRC_SCALAR = 0, 1, 2
NCOEF = 8
C_SQRT_AC, C_SQRT_1MAC, C_POST1, C_POST2, C_LOGVAR, C_RECIP, C_RECIPM1, C_SPARE = range(8)

vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double


class Layer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv", "bqkv", "wproj", "bproj",
                                  "w1", "b1", "w2", "b2", "ready")]


class Model(C.Structure):
    _fields_ = [(n, i32) for n in ("dtype", "d", "n_head", "n_layer", "T", "D", "nfreq", "n_scalar",
                                   "n_in_tok", "n_out_tok")] + \
               [(n, vp) for n in ("freqs", "pos_emb", "ln_in_w", "ln_in_b", "ln_f_w", "ln_f_b")] + \
               [("enc_w", vp * MAX_TOKENS), ("enc_w_ld", i64 * MAX_TOKENS), ("enc_k", i32 * MAX_TOKENS),
                ("enc_src", i32 * MAX_TOKENS), ("enc_off", i32 * MAX_TOKENS),
                ("dec_w", vp * MAX_TOKENS), ("dec_n", i32 * MAX_TOKENS), ("dec_off", i32 * MAX_TOKENS),
                ("layers", C.POINTER(Layer)), ("dec_ready", vp)]


class LayerGrad(C.Structure):
    _fields_ = [(n, vp) for n in ("wqkv_t", "wproj_t", "w1_t", "w2_t", "g_ln1_w", "g_ln1_b", "g_ln2_w", "g_ln2_b",
                                  "g_wqkv", "g_bqkv", "g_wproj", "g_bproj", "g_w1", "g_b1", "g_w2", "g_b2")]


class Backward(C.Structure):
    _fields_ = [("layers", C.POINTER(LayerGrad))] + \
               [(n, vp) for n in ("g_pos_emb", "g_ln_in_w", "g_ln_in_b", "g_ln_f_w", "g_ln_f_b")] + \
               [("g_enc_w", vp * MAX_TOKENS), ("g_dec_w", vp * MAX_TOKENS)] + \
               [(n, vp) for n in ("g", "gt", "d_h", "d_qkv", "d_act", "tA", "tB", "d_tok")] + \
               [("wgrad_overwrite", i32), ("wgrad_defer", i32), ("wgrad_sumsq", vp), ("ln_partials", vp), ("ln_partials_bytes", i64)]


class OptSegment(C.Structure):
    _fields_ = [("start", i64), ("count", i64), ("packed_off", i64), ("k", i32), ("ld", i32), ("decay", i32),
                ("reserved", i32)]


class PackItem(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("first4", i64), ("count", i64), ("k", i32), ("ld", i32), ("dst_dtype", i32),
                ("reserved", i32)]


class Workspace(C.Structure):
    _fields_ = [("B_max", i32), ("layer_stride", i32)] + \
               [(n, vp) for n in ("scal", "enc", "x", "h1", "qkv", "att", "h2", "pre", "act", "hf", "stats")]


_PROTOS = {
    "gpt_abi_version": ([], i32),
    "gpt_gemm_nt": ([i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp], i32),
    "gpt_gemm_nt_static": ([i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp], i32),
    "gpt_gemm_set_workspace": ([vp, C.c_size_t], i32),
    "gpt_gemm_tn": ([vp, i64, vp, i64, vp, i64, i32, i32, i32, i32, vp], i32),
    "gpt_gemm_tn_grouped": ([i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp], i32),
    "gpt_gemm_prof_enable": ([i32], i32),
    "gpt_gemm_prof_collect": ([C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)], i32),
    "gpt_gemm_resadd_layernorm": ([i32, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp], i32),
    "gpt_layernorm": ([vp, vp, vp, vp, i32, vp, i32, vp, vp, i32, i32, f32, vp], i32),
    "gpt_attention": ([i32, vp, vp, i32, i32, i32, i32, vp], i32),
    "gpt_scalar_tokens": ([vp, vp, vp, i32, vp, i32, i32, vp, i32, vp], i32),
    "gpt_encode_tokens": ([i32, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), C.POINTER(i32),
                           vp, i32, i32, vp], i32),
    "gpt_encode_tokens_subset": ([i32, i32, C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), C.POINTER(i32),
                                  C.POINTER(i32), i32, vp, i32, i32, vp], i32),
    "gpt_decode_tokens": ([i32, i32, vp, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), vp, vp, i32, i32, i32,
                           vp], i32),
    "gpt_q_sample": ([vp, vp, vp, vp, vp, i32, i32, vp], i32),
    "gpt_posterior_step": ([vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, i32, i32, vp], i32),
    "gpt_ddim_step": ([vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, f32, vp, vp, i32, i32, vp], i32),
    "gpt_mse_rows": ([vp, vp, vp, i32, i32, vp], i32),
    "gpt_mse_rows_bwd": ([vp, vp, vp, vp, i32, i32, vp], i32),
    "gpt_randn": ([vp, i64, u64, u64, vp], i32),
    "gpt_fill_i64": ([vp, i64, i64, vp], i32),
    "gpt_f32_to_bf16": ([vp, vp, i64, vp], i32),
    "gpt_bf16_to_f32": ([vp, vp, i64, vp], i32),
    "gpt_transpose": ([i32, vp, vp, i32, i32, i64, vp], i32),
    "gpt_layernorm_bwd": ([vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, vp], i32),
    "gpt_layernorm_bwd_colsum": ([vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp], i32),
    "gpt_attention_bwd": ([i32, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "gpt_colsum": ([i32, vp, i64, vp, i32, i32, vp], i32),
    "gpt_small_gemm": ([i32, vp, i64, i64, i32, vp, i64, i64, vp, i64, i32, i32, i32, i32, vp], i32),
    "gpt_small_gemm_batched": ([i32, i32, C.POINTER(vp), i64, i64, i32, C.POINTER(vp), i64, i64, C.POINTER(vp), C.POINTER(i32),
                                C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, vp], i32),
    "gpt_perm_index_mlp": ([vp, vp, i32, i32, i32, i32, vp], i32),
    "gpt_perm_index_cnn": ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    "gpt_perm_index_chain": ([vp, vp, i32, i32, i32, C.POINTER(i32), vp], i32),
    "gpt_perm_gather": ([C.POINTER(vp), C.POINTER(vp), i32, vp, i32, i32, vp], i32),
    "gpt_denoiser_forward": ([C.POINTER(Model), C.POINTER(Workspace), vp, vp, vp, vp, i32, vp, vp, i32, vp], i32),
    "gpt_denoiser_backward": ([C.POINTER(Model), C.POINTER(Workspace), C.POINTER(Backward), vp, vp, vp, i32, vp], i32),
    "gpt_denoiser_backward_stages": ([C.POINTER(Model), C.POINTER(Workspace), C.POINTER(Backward), vp, vp, vp, i32, i32, i32,
                                      vp], i32),
    "gpt_ddim_reverse_step": ([vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, i32, i32, vp], i32),
    "gpt_prior_bpd": ([vp, f32, f32, vp, i32, i32, vp], i32),
    "gpt_pair_gather": ([vp, i64, i64, vp, vp, vp, C.c_float, vp, vp, i32, i32, vp], i32),
    "gpt_cnn_eval": ([vp, vp, vp, i64, vp, i32, vp, vp, i32, i32, vp], i32),
    "gpt_mlp_head": ([vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "gpt_grad_sumsq": ([vp, i64, vp, vp, vp], i32),
    "gpt_adamw_ema_step": ([vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, vp, f32, f32, f64, f64, f64, f64, f64, i32, f64,
                            i32, vp], i32),
    "gpt_pack_segments": ([vp, vp, i32, vp, i32, i64, vp], i32),
    "gpt_pack_tensors": ([vp, i32, i64, vp], i32),
    "gpt_pack_tensors_if_changed": ([vp, i32, i64, vp, vp], i32),
    "gpt_dp_unique_id": ([vp], i32),
    "gpt_dp_init": ([vp, i32, i32], i32),
    "gpt_dp_destroy": ([], i32),
    "gpt_dp_rank": ([], i32),
    "gpt_dp_world": ([], i32),
    "gpt_dp_allreduce_grads": ([vp, i64, vp], i32),
    "gpt_dp_reduce_scatter": ([vp, vp, i64, i32, vp], i32),
    "gpt_dp_allgather": ([vp, vp, i64, i32, vp], i32),
    "gpt_dp_broadcast": ([vp, i64, i32, i32, vp], i32),
    "gpt_dp_barrier": ([vp], i32),
    "gpt_sample_loop": ([C.POINTER(Model), C.POINTER(Workspace), vp, vp, vp, i32, vp, vp, C.POINTER(i64), i32, i32,
                         i32, i32, i32, f32, f32, vp, u64, vp, vp, vp, vp, i32, vp], i32),
    "gpt_ddim_sample_loop": ([C.POINTER(Model), C.POINTER(Workspace), vp, vp, vp, i32, vp, vp, vp, f32, C.POINTER(i64), i32,
                              i32, i32, i32, i32, f32, f32, vp, u64, vp, vp, vp, vp, i32, vp], i32),
}

_lib = None


class GptHipError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GptHipError(f"{LIB_PATH} is missing: build it with `make -C g.pt_amd/csrc` "
                              "(there is no CPU fallback for the HIP path)")
        L = C.CDLL(LIB_PATH)
        L.gpt_last_error.restype = C.c_char_p
        L.gpt_last_error.argtypes = []
        for name, (args, res) in _PROTOS.items():
            fn = getattr(L, name)          # AttributeError here = the .so does not export what the header declares
            fn.argtypes, fn.restype = args, res
        _lib = L
    return _lib


def exported_symbols():
    return ["gpt_last_error"] + list(_PROTOS)


def check(status, what):
    if status != 0:
        raise GptHipError(f"{what} failed (status {status}): {lib().gpt_last_error().decode()}")


def ptr(t):
    """Device pointer of a GPU tensor (None -> NULL).  CPU tensors are refused: no silent fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GptHipError("the HIP path needs GPU tensors (got a CPU tensor); there is no CPU fallback")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dtype):
    if dtype in (torch.float32, "fp32", "f32", F32):
        return F32
    if dtype in (torch.bfloat16, "bf16", BF16):
        return BF16
    raise ValueError(f"unsupported compute dtype {dtype!r} (fp32 or bf16)")


def torch_dtype(code):
    return torch.bfloat16 if code == BF16 else torch.float32


_gemm_ws = {}


def ensure_gemm_workspace(device, nbytes=64 << 20):
    """Registers (once per device) the scratch gpt_gemm_nt uses to contract small-batch problems in K spans; the buffer
    lives as long as the process.  Users of the C-ABI alone call gpt_gemm_set_workspace themselves."""
    import torch
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _gemm_ws:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", idx))
        with torch.cuda.device(idx):
            check(lib().gpt_gemm_set_workspace(ptr(buf), nbytes), "gpt_gemm_set_workspace")
        _gemm_ws[idx] = buf
    return _gemm_ws[idx]
