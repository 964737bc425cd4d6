"""ctypes binding of libnnr_hip.so (include/nnr_hip.h).  No torch types cross the boundary: only raw device
pointers, sizes and the HIP stream handle.  The product path has NO fallback: if the library is missing or a call
fails, an exception is raised."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NNR_HIP_LIB') or os.path.join(_HERE, 'libnnr_hip.so')      # NNR_HIP_LIB: A/B a differently built library
_lib = None

vp, ci, cf, cu32, cl = C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_long


class GemmArgs(C.Structure):
    _fields_ = [('A', vp), ('B', vp), ('C', vp), ('M', ci), ('N', ci), ('K', ci), ('lda', ci), ('ldb', ci), ('ldc', ci),
                ('trans_a', ci), ('trans_b', ci), ('dyn_dev', vp), ('dyn_dim', ci), ('a_idx', vp), ('b_idx', vp),
                ('drop_target', ci), ('drop_p', cf), ('drop_seed', cu32), ('drop_cols', ci), ('alpha', cf), ('bias', vp),
                ('rowvec', vp), ('ldrv', ci), ('rowvec_map', vp), ('act', ci), ('aux_out', vp), ('ldaux', ci), ('mul', vp),
                ('ldmul', ci), ('resid', vp), ('ldres', ci), ('accumulate', ci), ('atomic', ci), ('c_idx', vp),
                ('split_k', ci), ('rowdot_w', vp), ('rowdot_out', vp), ('batch', ci), ('strideA', cl), ('strideB', cl),
                ('strideC', cl), ('stride_aux', cl), ('stride_res', cl), ('k_chunk', ci), ('colsum_out', vp), ('tile', ci), ('drop_thresh', cu32),
                ('drop_scale', cf), ('vec_epi', ci), ('sched', ci), ('slab', vp), ('slab_floats', cl), ('slab_mode', ci), ('pre_add', vp), ('ldpre', ci), ('gate_bwd', ci), ('B3', vp), ('b3_stride', cl), ('ldb3', ci)]


class LstmProblem(C.Structure):
    _fields_ = [('bs', vp), ('off', vp), ('slen', vp), ('prev_f', vp), ('prev_r', vp), ('n', ci), ('L', ci), ('gates', vp),
                ('cell', vp), ('hout', vp), ('cn', vp), ('wf', vp), ('wb', vp), ('dh', vp), ('dcn', vp), ('sync', vp)]


class TransposeDesc(C.Structure):
    _fields_ = [('inp', vp), ('out', vp), ('rows', ci), ('cols', ci)]


class CorpusTables(C.Structure):
    _fields_ = [(k, vp) for k in ('news_category', 'news_subCategory', 'title_text', 'title_mask', 'title_entity', 'abstract_text',
                                  'abstract_mask', 'abstract_entity', 'beh_user', 'beh_history', 'beh_history_mask', 'beh_line',
                                  'graph_table', 'cmask_table', 'cidx_table')] + [(k, ci) for k in ('T', 'C', 'H', 'G', 'K1')]


class BatchOut(C.Structure):
    _fields_ = [(k, vp) for k in ('user_id', 'u_cat', 'u_sub', 'u_tt', 'u_tm', 'u_te', 'u_ct', 'u_cm', 'u_ce', 'u_hmask', 'u_graph',
                                  'u_cmask', 'u_cidx', 'n_cat', 'n_sub', 'n_tt', 'n_tm', 'n_te', 'n_ct', 'n_cm', 'n_ce')]


class PoolArgs(C.Structure):
    _fields_ = [('x', vp), ('ldx', ci), ('D', ci), ('n', ci), ('L', ci), ('packed', ci), ('off', vp), ('slen', vp),
                ('order', vp), ('mask', vp), ('mask_div', ci), ('score', vp), ('v', vp), ('ldv', ci), ('scale', cf),
                ('alpha', vp), ('out', vp), ('ldo', ci), ('add_in', vp), ('ldadd', ci), ('dout', vp), ('lddo', ci),
                ('dout2', vp), ('lddo2', ci), ('dx', vp), ('lddx', ci), ('dx_accumulate', ci), ('dscore', vp), ('dv', vp),
                ('lddv', ci), ('th', vp), ('ldth', ci), ('A', ci), ('w2', vp),
                ('alpha_b', vp), ('dout_b', vp), ('lddo_b', ci), ('dscore_b', vp), ('v_b', vp), ('ldv_b', ci), ('scale_b', cf)]


# every symbol include/nnr_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    'nnr_version', 'nnr_gemm_f32', 'nnr_split_bf16x3', 'nnr_seq_plan', 'nnr_seq_plan_pair', 'nnr_cne_pair_map', 'nnr_lstm_dims', 'nnr_lstm_pack_weights', 'nnr_lstm_unpack_grads',
    'nnr_lstm_fwd', 'nnr_lstm_bwd', 'nnr_lstm_sync_bytes', 'nnr_lstm_sync_diag_offset', 'nnr_lstm_set_timeout_counter', 'nnr_attn_pool_fwd', 'nnr_attn_pool_bwd', 'nnr_gate_bwd', 'nnr_packed_seq_sum',
    'nnr_tanh_score_bwd', 'nnr_slot_workspace_floats', 'nnr_colsum', 'nnr_rowdot', 'nnr_small_embed_fwd', 'nnr_small_embed_bwd', 'nnr_add', 'nnr_add_atomic', 'nnr_add2d', 'nnr_expand_rows_fwd', 'nnr_expand_rows_bwd', 'nnr_dropout',
    'nnr_layernorm_fwd', 'nnr_layernorm_bwd', 'nnr_relu_bwd', 'nnr_relu_drop_bwd', 'nnr_gcn_aggregate_fwd', 'nnr_gcn_aggregate_bwd', 'nnr_sue_x0_fwd', 'nnr_sue_x0_bwd', 'nnr_sue_slice_fwd', 'nnr_sue_slice_bwd',
    'nnr_sue_intra_fwd', 'nnr_sue_intra_bwd', 'nnr_logits_loss_fwd', 'nnr_logits_fwd', 'nnr_nls_loss', 'nnr_logits_bwd', 'nnr_sumsq', 'nnr_sumsq_part', 'nnr_clip_adam',
    'nnr_mhsa_fwd', 'nnr_mhsa_bwd', 'nnr_mhsa_fwd_packed', 'nnr_mhsa_bwd_packed', 'nnr_mhsa_pair_map', 'nnr_mhsa_fwd_paired', 'nnr_mhsa_bwd_paired', 'nnr_mask_cover', 'nnr_seq_rowmap', 'nnr_embed_gather', 'nnr_embed_scatter', 'nnr_embed_scatter_dyn', 'nnr_transpose2d', 'nnr_transpose_batch', 'nnr_corpus_batch', 'nnr_history_graph', 'nnr_rank_metrics',
    'nnr_dp_unique_id', 'nnr_dp_init', 'nnr_dp_allreduce', 'nnr_dp_broadcast', 'nnr_dp_destroy', 'nnr_dp_emulate_ranks', 'nnr_dp_busy',
    'nnr_fill_zero', 'nnr_copy_bytes', 'nnr_fill_column_u8', 'nnr_adam_skipped_steps', 'nnr_adam_skipped_peek', 'nnr_fusion_rows_fwd', 'nnr_fusion_rows_bwd', 'nnr_click_loss',
    'nnr_tape_create', 'nnr_tape_destroy', 'nnr_tape_fn_id', 'nnr_tape_fn_nargs', 'nnr_tape_call', 'nnr_tape_wait_stream', 'nnr_tape_event_record',
    'nnr_tape_event_wait', 'nnr_tape_segment', 'nnr_tape_patch', 'nnr_tape_finalize', 'nnr_tape_info', 'nnr_tape_replay', 'nnr_tape_prepare_timing', 'nnr_tape_timings', 'nnr_tape_timeline',
    'nnr_tape_last_error',
    'nnr_token_sort_workspace_bytes', 'nnr_token_sort', 'nnr_embed_scatter_sorted_workspace_floats', 'nnr_embed_scatter_sorted', 'nnr_fusion_rows_bwd_det',
    'nnr_rows_touch', 'nnr_rows_compact', 'nnr_rows_pack', 'nnr_rows_unpack',
]


class NnrHipError(RuntimeError):
    pass


def build(force=False):
    """Compile libnnr_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        bdir = os.path.join(_HERE, 'csrc', 'build')
        if os.path.isdir(bdir):
            for f in os.listdir(bdir):
                os.remove(os.path.join(bdir, f))
    subprocess.check_call(['bash', os.path.join(_HERE, 'csrc', 'build.sh')])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NnrHipError('libnnr_hip.so not found at %s -- run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(there is no CPU / PyTorch fallback on the product path)' % LIB_PATH)
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be in the process BEFORE this library is
        # loaded, so that the library's DT_NEEDED entry binds to that copy.  Loaded first, libnnr_hip.so would pull in /opt/rocm's
        # runtime next to torch's: two HIP runtimes in one process, and every torch stream / allocation handed to an entry point is
        # foreign to the second one (each call fails with NNR_ERR_LAUNCH) -- found by running build() and smoke() in ONE process.
        import torch  # noqa: F401
        _lib = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(_lib, s).restype = ci
        _lib.nnr_lstm_sync_bytes.restype = C.c_size_t
        _lib.nnr_lstm_sync_diag_offset.restype = C.c_size_t
        _lib.nnr_token_sort_workspace_bytes.restype = C.c_size_t
        _lib.nnr_embed_scatter_sorted_workspace_floats.restype = C.c_size_t
    return _lib


def build_id():
    """Identity of the kernels that are running: sha256 over the sources libnnr_hip.so is built from (csrc/*.hip, common.h, the
    header) and, separately, over the loaded binary.  profiles/pmc_traffic.json carries both: counter traffic collected on another
    build is not quoted (nnr_amd.profile.pmc_traffic)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, 'csrc')
    for f in sorted(os.listdir(src)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(src, f), 'rb').read())
    h.update(open(os.path.join(os.path.dirname(_HERE), 'include', 'nnr_hip.h'), 'rb').read())
    lib_hash = hashlib.sha256(open(LIB_PATH, 'rb').read()).hexdigest()[:16] if os.path.exists(LIB_PATH) else None
    return {'src_sha256': h.hexdigest()[:16], 'lib_sha256': lib_hash}


CALLS = [0]          # C-ABI calls checked so far (bench.py reports calls per step; each is one or a few kernel launches)


def check(rc, what):
    CALLS[0] += 1
    if rc != 0:
        raise NnrHipError('%s failed with code %d' % (what, rc))
