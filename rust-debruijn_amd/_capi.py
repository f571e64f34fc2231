"""ctypes binding of include/dbg_mi355x.h (the C-ABI drop-in boundary).  No torch types cross it."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBG_LIB") or os.path.join(HERE, "libdbg_mi355x.so")   # DBG_LIB: an experimental build, for measurements

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)
u8p = C.POINTER(C.c_uint8)


class SeqSet(C.Structure):
    _fields_ = [("words", C.c_void_p), ("n_words", C.c_uint64), ("start", C.c_void_p), ("length", C.c_void_p),
                ("exts", C.c_void_p), ("data", C.c_void_p), ("data_width", C.c_uint32), ("n_seqs", C.c_uint64)]


class FilterParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("stranded", C.c_int32), ("summarizer", C.c_int32), ("min_kmer_obs", C.c_uint64),
                ("report_all_kmers", C.c_int32), ("memory_size", C.c_uint64), ("compact_sets", C.c_uint32)]


class KmerTable(C.Structure):
    _fields_ = [("n", C.c_uint64), ("key_hi", C.c_void_p), ("key_lo", C.c_void_p), ("exts", C.c_void_p),
                ("count", C.c_void_p), ("set_off", C.c_void_p), ("set_val", C.c_void_p), ("n_set_val", C.c_uint64),
                ("n_all", C.c_uint64), ("all_hi", C.c_void_p), ("all_lo", C.c_void_p),
                ("n_kmer_instances", C.c_uint64), ("n_passes", C.c_uint32), ("on_device", C.c_int32),
                ("set_off_width", C.c_uint32), ("set_val_width", C.c_uint32)]


class MspParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("p", C.c_uint32), ("permutation", C.c_void_p), ("rc", C.c_int32),
                ("lmer_words", C.c_uint32)]


class MspPieces(C.Structure):
    _fields_ = [("n_pieces", C.c_uint64), ("piece_off", C.c_void_p), ("bucket", C.c_void_p), ("exts", C.c_void_p),
                ("start", C.c_void_p), ("len", C.c_void_p), ("minimizer_pos", C.c_void_p), ("lmer", C.c_void_p),
                ("on_device", C.c_int32)]


class Graph(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("seq_words", C.c_void_p), ("n_seq_words", C.c_uint64),
                ("seq_len_bases", C.c_uint64), ("start", C.c_void_p), ("length", C.c_void_p), ("exts", C.c_void_p),
                ("data", C.c_void_p), ("stranded", C.c_int32)]


class LabelClasses(C.Structure):
    _fields_ = [("n_classes", C.c_uint64), ("set_off", C.c_void_p), ("set_val", C.c_void_p), ("n_set_val", C.c_uint64)]


class Edges(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("target", C.c_void_p), ("info", C.c_void_p)]


class SynthParams(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("read_len", C.c_uint32), ("genome_len", C.c_uint64),
                ("genome_seed", C.c_uint64), ("read_seed", C.c_uint64), ("error_rate", C.c_double),
                ("stranded", C.c_int32), ("n_colours", C.c_uint32), ("first_read", C.c_uint64)]


class ShardPlan(C.Structure):
    _fields_ = [("k", C.c_uint32), ("stranded", C.c_int32), ("summarizer", C.c_int32), ("min_kmer_obs", C.c_uint64),
                ("total_kmers", C.c_uint64), ("n_bins", C.c_uint32), ("rec_words", C.c_uint32), ("bin_group", C.c_uint32),
                ("max_label", C.c_uint32), ("merge_dups", C.c_uint32), ("n_labels", C.c_uint32), ("labels", C.c_uint32 * 64)]


# ---- the rank-spanning flow (dbg_transport, dbg_shard_filter_kmers_dev, dbg_shard_compress_dev) ----
TR_ALL_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p)
TR_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
TR_ALL_TO_ALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, u64p, u64p, C.c_void_p, u64p, u64p, C.c_void_p)
TR_SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p)
TR_RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p)
TR_POLL = C.CFUNCTYPE(C.c_int, C.c_void_p)
TR_ABORT = C.CFUNCTYPE(None, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("self", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("all_reduce_u64", TR_ALL_REDUCE),
                ("all_gather", TR_ALL_GATHER), ("all_to_allv", TR_ALL_TO_ALLV), ("send", TR_SEND), ("recv", TR_RECV),
                ("poll", TR_POLL), ("abort", TR_ABORT)]


class ShardParams(C.Structure):
    _fields_ = [("k", C.c_uint32), ("stranded", C.c_int32), ("summarizer", C.c_int32), ("min_kmer_obs", C.c_uint64),
                ("n_rounds", C.c_uint32), ("merge_dups", C.c_int32), ("balance", C.c_int32), ("force_exchange", C.c_int32)]


class ShardStats(C.Structure):
    _fields_ = [("total_kmers", C.c_uint64), ("local_kmers", C.c_uint64), ("records_scanned", C.c_uint64),
                ("records_owned", C.c_uint64), ("bytes_sent", C.c_uint64), ("n_bins", C.c_uint32), ("owned_lo", C.c_uint32),
                ("owned_hi", C.c_uint32), ("n_rounds", C.c_uint32), ("merge_dups", C.c_int32), ("balanced", C.c_int32),
                ("exposed_ms", C.c_double), ("exposed_ms_round", C.c_double * 64), ("setup_ms", C.c_double)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_double), ("launches", C.c_uint32), ("units", C.c_uint64)]


class CtxStats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("slab_backing", C.c_uint32), ("slab_bytes", C.c_uint64),                 ("slab_rec_words", C.c_uint32), ("slab_pooled", C.c_uint32), ("pooled_bytes", C.c_uint64), ("pooled_high_water", C.c_uint64),
                ("n_hipmalloc", C.c_uint64), ("n_fresh_blocks", C.c_uint64), ("n_pool_hits", C.c_uint64), ("n_trims", C.c_uint64),
                ("n_oom_retries", C.c_uint64), ("n_raw_free", C.c_uint64), ("n_pinned_alloc", C.c_uint64),
                ("s_hipmalloc", C.c_double), ("s_free", C.c_double), ("s_pinned_alloc", C.c_double),
                ("slab_note", C.c_char * 96), ("slab_trials_done", C.c_uint32), ("slab_candidates_pooled", C.c_uint32),
                ("slab_trial_ms", C.c_float * 8)]


SLAB_BACKING_NAMES = {0: "none", 1: "plain block, tournament", 2: "plain (slab < 4 GB)", 5: "no slabs (read-order buffer + scatter)"}


# every symbol include/dbg_mi355x.h declares
EXPORTS = [
    "dbg_ctx_create", "dbg_ctx_destroy", "dbg_last_error", "dbg_version", "dbg_ctx_set_stream",
    "dbg_ctx_set_scratch_budget", "dbg_ctx_set_option", "dbg_ctx_trim", "dbg_seqset_max_label_dev", "dbg_seqset_label_bitmap_dev", "dbg_seqset_to_device", "dbg_seqset_free_device", "dbg_filter_kmers", "dbg_filter_kmers_dev", "dbg_free_table", "dbg_table_to_host",
    "dbg_remove_censored_exts", "dbg_msp_sequence", "dbg_msp_sequence_dev", "dbg_free_pieces",
    "dbg_compress_kmers_with_hash", "dbg_compress_kmers_with_hash_dev", "dbg_kmer_set_exts", "dbg_compress_kmers_no_exts", "dbg_free_graph", "dbg_label_classes_dev", "dbg_free_label_classes", "dbg_compress_table_dev", "dbg_synth_words", "dbg_synth_reads_dev",
    "dbg_synth_reads_host", "dbg_ctx_enable_timing", "dbg_ctx_get_timings",
    "dbg_ctx_get_stats", "dbg_ctx_probe_slab", "dbg_ctx_warm", "dbg_abi_version",
    "dbg_count_kmer_instances_dev", "dbg_shard_plan_make", "dbg_shard_scan_dev", "dbg_shard_scatter_dev",
    "dbg_shard_count_dev", "dbg_shard_count_begin", "dbg_shard_count_bins_dev", "dbg_shard_count_finish", "dbg_graph_combine", "dbg_compress_graph",
    "dbg_graph_edges", "dbg_free_edges", "dbg_graph_to_gfa", "dbg_graph_write_gfa", "dbg_free_text",
    "dbg_graph_serialize", "dbg_graph_deserialize", "dbg_free_bytes", "dbg_serde_last_error",
    "dbg_transport_rccl_create", "dbg_transport_destroy", "dbg_transport_aborted", "dbg_transport_inprocess_create", "dbg_rccl_unique_id", "dbg_rccl_comm_create", "dbg_rccl_comm_destroy",
    "dbg_shard_owner_bounds", "dbg_shard_round_cuts", "dbg_shard_filter_kmers_dev", "dbg_shard_compress_dev",
    "dbg_pack_acgt", "dbg_pack_acgt_dev", "dbg_pack_acgt_hashn", "dbg_pack_acgt_hashn_dev", "dbg_unpack_acgt", "dbg_unpack_acgt_dev",
]

_lib = None


def load():
    """Load the HIP library.  There is no CPU fallback: a missing library is a hard error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libdbg_mi355x.so is not built (run `python __graft_entry__.py` / _build.build()); "
                          "the MI355X hot path has no CPU fallback")
    # torch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both drive the
    # GPU, so let torch's copy load first and satisfy this library's DT_NEEDED by SONAME.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.dbg_last_error.restype = C.c_char_p
    lib.dbg_last_error.argtypes = [C.c_void_p]
    lib.dbg_version.restype = C.c_char_p
    lib.dbg_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.dbg_ctx_destroy.argtypes = [C.c_void_p]
    lib.dbg_ctx_destroy.restype = None
    lib.dbg_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.dbg_ctx_set_scratch_budget.argtypes = [C.c_void_p, C.c_uint64]
    lib.dbg_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    lib.dbg_ctx_trim.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.dbg_seqset_max_label_dev.argtypes = [C.c_void_p, C.POINTER(SeqSet), C.POINTER(C.c_uint32)]
    lib.dbg_seqset_label_bitmap_dev.argtypes = [C.c_void_p, C.POINTER(SeqSet), C.POINTER(C.c_uint32)]
    lib.dbg_seqset_to_device.argtypes = [C.c_void_p, C.POINTER(SeqSet), C.POINTER(SeqSet)]
    lib.dbg_seqset_free_device.argtypes = [C.c_void_p, C.POINTER(SeqSet)]
    lib.dbg_seqset_free_device.restype = None
    lib.dbg_filter_kmers.argtypes = [C.c_void_p, C.POINTER(SeqSet), C.POINTER(FilterParams), C.POINTER(KmerTable)]
    lib.dbg_filter_kmers_dev.argtypes = lib.dbg_filter_kmers.argtypes
    lib.dbg_free_table.argtypes = [C.c_void_p, C.POINTER(KmerTable)]
    lib.dbg_free_table.restype = None
    lib.dbg_table_to_host.argtypes = [C.c_void_p, C.POINTER(KmerTable), C.POINTER(KmerTable)]
    lib.dbg_remove_censored_exts.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(KmerTable), C.c_int]
    lib.dbg_msp_sequence.argtypes = [C.c_void_p, C.POINTER(SeqSet), C.POINTER(MspParams), C.POINTER(MspPieces)]
    lib.dbg_msp_sequence_dev.argtypes = lib.dbg_msp_sequence.argtypes
    lib.dbg_free_pieces.argtypes = [C.c_void_p, C.POINTER(MspPieces)]
    lib.dbg_free_pieces.restype = None
    lib.dbg_compress_kmers_with_hash.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Graph)]
    lib.dbg_compress_kmers_with_hash_dev.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Graph)]
    lib.dbg_kmer_set_exts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dbg_compress_kmers_no_exts.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.POINTER(Graph)]
    lib.dbg_free_graph.argtypes = [C.c_void_p, C.POINTER(Graph)]
    lib.dbg_free_graph.restype = None
    lib.dbg_label_classes_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(LabelClasses)]
    lib.dbg_free_label_classes.argtypes = [C.POINTER(LabelClasses)]
    lib.dbg_free_label_classes.restype = None
    lib.dbg_compress_table_dev.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(KmerTable), C.POINTER(Graph),
                                           C.POINTER(LabelClasses)]
    lib.dbg_graph_combine.argtypes = [C.c_void_p, C.POINTER(Graph), C.c_uint32, C.POINTER(Graph)]
    lib.dbg_compress_graph.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(Graph), C.c_void_p, C.c_uint64,
                                       C.POINTER(Graph)]
    lib.dbg_graph_edges.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Graph), C.POINTER(Edges)]
    lib.dbg_free_edges.argtypes = [C.POINTER(Edges)]
    lib.dbg_free_edges.restype = None
    lib.dbg_graph_to_gfa.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Graph), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.dbg_graph_write_gfa.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Graph), C.c_char_p]
    lib.dbg_free_text.argtypes = [C.c_void_p]
    lib.dbg_free_text.restype = None
    lib.dbg_graph_serialize.argtypes = [C.c_void_p, C.POINTER(Graph), C.c_int, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    lib.dbg_graph_deserialize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(Graph)]
    lib.dbg_free_bytes.argtypes = [C.c_void_p]
    lib.dbg_free_bytes.restype = None
    lib.dbg_serde_last_error.restype = C.c_char_p
    lib.dbg_pack_acgt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.dbg_pack_acgt_dev.argtypes = lib.dbg_pack_acgt.argtypes
    lib.dbg_pack_acgt_hashn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_uint64)]
    lib.dbg_pack_acgt_hashn_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.POINTER(C.c_uint64)]
    lib.dbg_unpack_acgt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.dbg_unpack_acgt_dev.argtypes = lib.dbg_unpack_acgt.argtypes
    lib.dbg_synth_words.argtypes = [C.POINTER(SynthParams)]
    lib.dbg_synth_words.restype = C.c_uint64
    lib.dbg_synth_reads_dev.argtypes = [C.c_void_p, C.POINTER(SynthParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    lib.dbg_synth_reads_host.argtypes = [C.POINTER(SynthParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dbg_count_kmer_instances_dev.argtypes = [C.c_void_p, C.POINTER(SeqSet), C.c_uint32, C.POINTER(C.c_uint64)]
    lib.dbg_shard_plan_make.argtypes = [C.c_void_p, C.POINTER(ShardPlan)]
    lib.dbg_shard_scan_dev.argtypes = [C.c_void_p, C.POINTER(SeqSet), C.POINTER(ShardPlan), C.POINTER(C.c_uint64), C.c_void_p]
    lib.dbg_shard_scatter_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dbg_shard_count_dev.argtypes = [C.c_void_p, C.POINTER(ShardPlan), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                        C.c_uint64, C.POINTER(KmerTable)]
    lib.dbg_shard_count_begin.argtypes = [C.c_void_p, C.POINTER(ShardPlan), C.c_uint64]
    lib.dbg_shard_count_bins_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    lib.dbg_shard_count_finish.argtypes = [C.c_void_p, C.POINTER(KmerTable)]
    lib.dbg_transport_rccl_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.POINTER(Transport)), C.c_char_p, C.c_uint64]
    lib.dbg_transport_inprocess_create.argtypes = [C.c_int32, C.POINTER(C.POINTER(Transport))]
    lib.dbg_transport_destroy.argtypes = [C.POINTER(Transport)]
    lib.dbg_transport_destroy.restype = None
    lib.dbg_transport_aborted.argtypes = [C.POINTER(Transport)]
    lib.dbg_rccl_unique_id.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_uint64]
    lib.dbg_rccl_comm_create.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.c_char_p, C.c_uint64]
    lib.dbg_rccl_comm_destroy.argtypes = [C.c_char_p, C.c_void_p]
    lib.dbg_shard_owner_bounds.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.dbg_shard_round_cuts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
    lib.dbg_shard_filter_kmers_dev.argtypes = [C.c_void_p, C.POINTER(Transport), C.POINTER(SeqSet), C.POINTER(ShardParams),
                                               C.POINTER(KmerTable), C.POINTER(ShardStats)]
    lib.dbg_shard_compress_dev.argtypes = [C.c_void_p, C.POINTER(Transport), C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(KmerTable),
                                           C.c_int32, C.c_int32, C.POINTER(Graph), C.POINTER(Graph), C.POINTER(LabelClasses)]
    lib.dbg_ctx_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.dbg_ctx_get_timings.argtypes = [C.c_void_p, C.POINTER(KernelTime), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.dbg_ctx_get_stats.argtypes = [C.c_void_p, C.POINTER(CtxStats)]
    lib.dbg_ctx_probe_slab.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    lib.dbg_ctx_warm.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    lib.dbg_abi_version.restype = C.c_uint32
    lib.dbg_abi_version.argtypes = []
    _lib = lib
    return lib
