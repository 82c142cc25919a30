"""ctypes binding of libmpmae_hip.so (C ABI declared in include/mpmae_hip.h).

The product path has NO CPU / PyTorch fallback: if the HIP library is missing or an entry
point fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPMAE_LIB: developer override for A/B runs of two builds inside one GPU session
LIB_PATH = os.environ.get("MPMAE_LIB") or os.path.join(_HERE, "libmpmae_hip.so")

c_void_p, c_int, c_float, c_size_t = C.c_void_p, C.c_int, C.c_float, C.c_size_t


class Geom(C.Structure):
    _fields_ = [("vis", c_void_p), ("inv", c_void_p),
                ("N", c_int), ("keep", c_int), ("grid", c_int), ("S", c_int)]


class GemmArgs(C.Structure):
    _fields_ = [("A", c_void_p), ("A2", c_void_p), ("B", c_void_p), ("bias", c_void_p),
                ("C", c_void_p), ("R", c_void_p),
                ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int),
                ("ldc", c_int), ("ldr", c_int),
                ("p0", c_void_p), ("p1", c_void_p),
                ("rpg", c_int),
                ("s0", c_void_p), ("s1", c_void_p),
                ("vis", c_void_p), ("inv", c_void_p), ("act", c_void_p), ("act_src", c_void_p),
                ("keep", c_int), ("L", c_int), ("S", c_int), ("Cseg", c_int), ("grid", c_int),
                ("H", c_int),
                ("ws", c_void_p), ("ws_floats", c_size_t)]


class WgradArgs(C.Structure):
    _fields_ = [("P", c_void_p), ("P2", c_void_p), ("Q", c_void_p),
                ("M", c_int), ("Nn", c_int), ("Kk", c_int), ("ldp", c_int), ("ldq", c_int),
                ("dW", c_void_p), ("sn", c_int), ("sk", c_int), ("db", c_void_p),
                ("pp0", c_void_p), ("pp1", c_void_p), ("qp0", c_void_p), ("qp1", c_void_p),
                ("rpg", c_int), ("rows_per_split", c_int),
                ("vis", c_void_p), ("inv", c_void_p), ("act_src", c_void_p),
                ("keep", c_int), ("L", c_int), ("S", c_int), ("Cseg", c_int), ("grid", c_int),
                ("H", c_int),
                ("ws", c_void_p), ("ws_floats", c_size_t), ("rowscale", c_void_p)]


class DwArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("add", c_void_p),
                ("w", c_void_p), ("bias", c_void_p),
                ("s_kh", c_int), ("s_kw", c_int), ("s_c", c_int), ("flip", c_int),
                ("g", Geom),
                ("C", c_int), ("CC", c_int), ("TP", c_int), ("tiles_side", c_int),
                ("act", c_void_p)]


class DwWgArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("dd", c_void_p), ("dw", c_void_p), ("db", c_void_p),
                ("s_kh", c_int), ("s_kw", c_int), ("s_c", c_int),
                ("g", Geom),
                ("C", c_int), ("CC", c_int), ("TP", c_int), ("tiles_side", c_int),
                ("ntiles_total", c_int),
                ("act", c_void_p),
                ("ws", c_void_p), ("ws_floats", c_size_t)]


class PrepDesc(C.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p),
                ("rows", c_int), ("cols", c_int), ("sr", c_int), ("sc", c_int),
                ("dst_ld", c_int), ("pad", c_int)]


class PixContArgs(C.Structure):
    _fields_ = [("pred", c_void_p), ("dpred", c_void_p), ("ld", c_int), ("coff", c_int),
                ("target", c_void_p), ("mask", c_void_p),
                ("C", c_int), ("p", c_int), ("grid", c_int), ("H", c_int), ("L", c_int),
                ("norm_pix", c_int),
                ("acc", c_void_p), ("patch_l", c_void_p), ("patch_cnt", c_void_p),
                ("patch_mean", c_void_p), ("patch_rstd", c_void_p),
                ("coef", c_void_p)]


class PixCatArgs(C.Structure):
    _fields_ = [("pred", c_void_p), ("dpred", c_void_p), ("ld", c_int), ("coff", c_int),
                ("target", c_void_p), ("mask", c_void_p),
                ("K", c_int), ("p", c_int), ("grid", c_int), ("H", c_int), ("L", c_int),
                ("acc", c_void_p), ("coef", c_void_p)]


class ImgArgs(C.Structure):
    _fields_ = [("pred", c_void_p), ("dpred", c_void_p), ("ld", c_int), ("coff", c_int),
                ("target", c_void_p),
                ("K", c_int), ("N", c_int), ("kind", c_int),
                ("acc", c_void_p), ("coef", c_void_p)]


class RsArgs(C.Structure):
    _fields_ = [("A", c_void_p), ("A2", c_void_p), ("W", c_void_p), ("ldw", c_int),
                ("bias", c_void_p), ("v0", c_void_p), ("v1", c_void_p),
                ("out", c_void_p), ("xhat", c_void_p), ("xn", c_void_p), ("rstd", c_void_p), ("R", c_void_p),
                ("lng", c_void_p), ("ws", c_void_p), ("act", c_void_p), ("M", c_int),
                ("C", c_int), ("H", c_int), ("s0", c_void_p), ("s1", c_void_p), ("ws_floats", c_size_t),
                ("rpg", c_int),
                ("fin_sum", c_void_p), ("fin_sum0", c_void_p), ("fin_gamma", c_void_p), ("fin_gx", c_void_p),
                ("fin_ainv", c_void_p), ("fin_out", c_void_p), ("fin_dgamma", c_void_p), ("fin_dbeta", c_void_p),
                ("fin_eps", C.c_float), ("dz_dout", c_void_p), ("dz_w2t", c_void_p), ("dz_ldw2", c_int), ("dz_bias", c_void_p),
                ("defer_fold", c_void_p), ("ln_done", c_int),
                ("wg_ws", c_void_p), ("wg_ws_floats", c_size_t), ("wg_rows", c_void_p),
                ("dn_xhat", c_void_p), ("dn_rstd", c_void_p), ("dn_y", c_void_p), ("dn_gamma", c_void_p), ("dn_beta", c_void_p), ("dn_S", c_int)]


class FoldDesc(C.Structure):
    _fields_ = [("part", c_void_p), ("P", c_int), ("W", c_int), ("out", c_void_p), ("a", c_int), ("b", c_int), ("c", c_int)]


PS_MAXBLK = 9          # MPMAE_PS_MAXBLK
PS_SYNC_WORDS = 640    # MPMAE_PS_SYNC_WORDS
DWG_MAX = 12           # problems per mpmae_dwconv7_wgrad_group launch (csrc/dwconv.cuh)
TNG_MAXP = 20          # problems per mpmae_wgrad_group launch (csrc/gemm_tng.cuh)


class PsBlock(C.Structure):
    _fields_ = [("dw_w", c_void_p), ("dw_b", c_void_p), ("ln_g", c_void_p), ("ln_b", c_void_p),
                ("W1", c_void_p), ("b1", c_void_p), ("grn_g", c_void_p), ("grn_b", c_void_p),
                ("W2", c_void_p), ("b2", c_void_p), ("ldw1", c_int), ("ldw2", c_int),
                ("dhat", c_void_p), ("rstd", c_void_p), ("xn", c_void_p), ("h", c_void_p), ("z", c_void_p), ("out", c_void_p),
                ("G2", c_void_p), ("Gx", c_void_p), ("Ainv", c_void_p), ("scale", c_void_p)]


class PsArgs(C.Structure):
    _fields_ = [("x_in", c_void_p), ("g", Geom), ("act", c_void_p),
                ("C", c_int), ("nblk", c_int), ("eps", c_float), ("ng", c_int),
                ("sync", c_void_p), ("sync_words", c_int), ("pad_", c_int), ("blk", PsBlock * PS_MAXBLK)]


class Meters(C.Structure):
    _fields_ = [("losses", c_void_p), ("weighted", c_void_p), ("T", c_int), ("ring", c_void_p), ("window", c_int),
                ("sums", c_void_p), ("gnorm2", c_void_p), ("err_words", c_void_p), ("n_err", c_int), ("err_stride", c_int)]


class StemTailArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("xhat1", c_void_p), ("rstd1", c_void_p), ("xhat2", c_void_p), ("rstd2", c_void_p),
                ("out", c_void_p), ("g1", c_void_p), ("b1", c_void_p), ("w", c_void_p), ("wb", c_void_p),
                ("g2", c_void_p), ("b2", c_void_p), ("act_in", c_void_p), ("act_out", c_void_p),
                ("dg1", c_void_p), ("db1", c_void_p), ("dw", c_void_p), ("dwb", c_void_p), ("dg2", c_void_p),
                ("db2", c_void_p), ("ws", c_void_p), ("ws_floats", c_size_t), ("M", c_int), ("C", c_int)]


class StemFrontArgs(C.Structure):
    _fields_ = [("img", c_void_p), ("vis", c_void_p), ("inv", c_void_p), ("W", c_void_p), ("ldw", c_int), ("W_master", c_void_p), ("bias", c_void_p),
                ("xhat1", c_void_p), ("rstd1", c_void_p), ("xhat2", c_void_p), ("rstd2", c_void_p), ("out", c_void_p),
                ("g1", c_void_p), ("b1", c_void_p), ("w", c_void_p), ("wb", c_void_p), ("g2", c_void_p), ("b2", c_void_p),
                ("col", c_void_p), ("ldc", c_int),
                ("N", c_int), ("keep", c_int), ("grid", c_int), ("H", c_int), ("Cin", c_int), ("C0", c_int),
                ("track_activity", c_int), ("act_out", c_void_p)]


OPT = {n: i for i, n in enumerate("DW DWW NT_GLDS64 NT_BK32 NT_GLDS TN CS_SPLIT RSC_PF RSC_N40 RSC_N80 TN3_BLOCKS TNG_BLOCKS FOLD_GROUP RSC_W5 RSC_ATOMIC DET RSC1 RSC1_ATOMIC RSP RSP_NWV RSP_NARROW RSN3 EVX RST_NW NT_RING".split())}      # enum MpmaeOption (include/mpmae_hip.h)
PRO = dict(NONE=0, LN_AFFINE=1, GRN=2, GRN_BWD=3, DOWN_GATHER=4, ROW_GATHER=5, IM2COL3=6)
EPI = dict(STORE=0, GELU_SUMSQ=1, RESID=2, DZ_STATS=3, SCATTER_ROWS=4, DOWN_DGRAD=5)

# every symbol include/mpmae_hip.h declares: name -> argtypes
P = C.POINTER
SYMBOLS = {
    "mpmae_arch": [],
    "mpmae_crop": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mpmae_crop_norm": [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p],
    "mpmae_crop_lut": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "mpmae_gather_kxk": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mpmae_mask_gen": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "mpmae_mask_gen_dense": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p],
    "mpmae_activity": [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mpmae_activity_pool": [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mpmae_prep_weights": [c_int, c_void_p, c_int, c_int, c_void_p],
    "mpmae_gemm": [c_int, c_int, c_int, P(GemmArgs), c_void_p],
    "mpmae_wgrad": [c_int, c_int, c_int, P(WgradArgs), c_int, c_void_p],
    "mpmae_fold_group": [P(FoldDesc), c_int, c_void_p],
    "mpmae_wgrad_group": [c_int, P(WgradArgs), c_int, c_void_p, c_size_t, c_void_p],
    "mpmae_ln_fwd": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                     c_int, c_int, c_void_p, c_void_p],
    "mpmae_ln_bwd": [c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                     c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p],
    "mpmae_ln_bwd_defer": [c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                           c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p],
    "mpmae_ln_bwd_down_defer": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_size_t, c_void_p, c_void_p],
    "mpmae_grn_fwd_finalize": [c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "mpmae_grn_bwd_finalize": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                               c_void_p, c_void_p, c_void_p],
    "mpmae_grn_apply_fin": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p],
    "mpmae_grn_bwd_apply_fin": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p],
    "mpmae_grn_stats_from_wgrad": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_void_p],
    "mpmae_grn_apply": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p],
    "mpmae_grn_bwd_apply": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mpmae_grn_group_ok": [c_int, c_int, c_int, c_int],
    "mpmae_grn_group_fwd": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p,
                            c_void_p, c_void_p],
    "mpmae_grn_group_bwd": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                            c_void_p],
    "mpmae_colstats": [c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                       c_void_p],
    "mpmae_rs": [c_int, P(RsArgs), c_void_p],
    "mpmae_rs_wgrad_fold": [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "mpmae_ps_fwd": [P(PsArgs), c_void_p],
    "mpmae_quant_mx": [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p],
    "mpmae_gemm_mx": [c_int, P(GemmArgs), c_void_p, c_int, c_void_p, c_int, c_void_p],
    "mpmae_set_option": [c_int, c_int],
    "mpmae_get_option": [c_int],
    "mpmae_dwconv7_fwd": [c_int, P(DwArgs), c_void_p],
    "mpmae_dwconv7_wgrad": [c_int, P(DwWgArgs), c_int, c_void_p],
    "mpmae_dwconv7_wgrad_group": [c_int, P(DwWgArgs), c_int, c_void_p, c_size_t, c_void_p],
    "mpmae_dwstride_fwd": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                           c_void_p, c_void_p, c_void_p],
    "mpmae_dwstride_bwd": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                           c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p],
    "mpmae_fill_mask_token": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "mpmae_mask_token_bwd": [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "mpmae_pool_rows": [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    "mpmae_loss_pix_cont": [c_int, c_int, P(PixContArgs), c_int, c_void_p],
    "mpmae_loss_pix_cat": [c_int, c_int, P(PixCatArgs), c_int, c_void_p],
    "mpmae_loss_img": [c_int, c_int, P(ImgArgs), c_void_p],
    "mpmae_loss_multi": [c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p],
    "mpmae_loss_pix_cont_rows": [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mpmae_loss_pix_cont_rows_bwd": [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mpmae_loss_pix_cont_rows_fused": [c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mpmae_head_scale": [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p],
    "mpmae_loss_pix_cat_waves": [c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p],
    "mpmae_loss_finalize": [c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p],
    "mpmae_loss_finalize_guarded": [c_void_p, c_int, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_void_p],
    "mpmae_adamw": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                    c_size_t, c_void_p, c_void_p, c_void_p],
    "mpmae_sumsq": [c_void_p, c_size_t, c_void_p, c_void_p],
    "mpmae_ln_fwd_down": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int,
                          c_void_p, c_void_p],
    "mpmae_ln_bwd_down": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                          c_void_p, c_void_p, c_size_t, c_void_p],
    "mpmae_im2col3": [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                      c_void_p],
    "mpmae_strided_add": [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p],
    "mpmae_stem_tail": [c_int, c_int, P(StemTailArgs), c_void_p],
    "mpmae_stem_front": [P(StemFrontArgs), c_void_p],
    "mpmae_hp_fetch": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, P(Meters), c_void_p],
    "mpmae_program_begin_op": [c_void_p, c_int, C.POINTER(c_int), c_int, c_int],
    "mpmae_program_end": [c_void_p],
    "mpmae_program_num_ops": [c_void_p],
    "mpmae_program_export_signal": [c_void_p, c_int],
    "mpmae_program_stream_wait": [c_void_p, c_int, c_void_p],
    "mpmae_program_run": [c_void_p, c_int, c_int, c_void_p],
    "mpmae_program_stream_overlaps": [c_void_p, c_void_p, c_void_p],
    "mpmae_memset_async": [c_void_p, c_int, c_size_t, c_void_p],
    "mpmae_memcpy_h2d_async": [c_void_p, c_void_p, c_size_t, c_void_p],
}
# entry points that do not return an error code: name -> (argtypes, restype)
OTHER_SYMBOLS = {
    "mpmae_program_create": ([], c_void_p),
    "mpmae_program_destroy": ([c_void_p], None),
}

_lib = None
OPT_DEFAULTS = {}            # filled by load(): the library's built-in option values


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load libmpmae_hip.so; raises HipLibraryError if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    # The library shares streams and device pointers with PyTorch, so both must sit on ONE HIP runtime (and one hipBLASLt): PyTorch ships
    # its own copies under torch/lib with the same sonames as ROCm's - whichever is loaded first serves both. Import torch first so that
    # the order (and therefore the copy) is always the same, whatever the caller imported before.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, (argtypes, restype) in OTHER_SYMBOLS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.mpmae_arch() != 950:
        raise HipLibraryError("libmpmae_hip.so was not built for gfx950")
    OPT_DEFAULTS.update({n: int(lib.mpmae_get_option(i)) for n, i in OPT.items()})      # a fresh library: every switch at its built-in default
    _lib = lib
    return lib


def nondefault_options():
    """Library switches (mpmae_set_option) that differ from the built-in defaults right now: {} on a clean process. bench.py prints them."""
    lib = load()
    return {n: int(lib.mpmae_get_option(i)) for n, i in OPT.items() if int(lib.mpmae_get_option(i)) != OPT_DEFAULTS[n]}


def check(err: int, what: str):
    if err != 0:
        raise HipLibraryError(f"{what} failed with hipError {err}")
