"""ctypes bindings for include/hiphase_gpu.h (the drop-in C ABI).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc --offload-arch=gfx950) as
``hiphase_amd/libhiphase_gpu.so``. Loading fails loudly when it is missing — there is no fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HP_LIB names another in-tree build of the same library (e.g. the instrumented one scripts/prof_wfa2.sh makes)
LIB_PATH = os.environ.get("HP_LIB") or os.path.join(_HERE, "libhiphase_gpu.so")


class HpError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"hiphase_gpu status {code}: {msg}")
        self.code = code


class BlockView(C.Structure):
    _fields_ = [
        ("n_variants", C.c_uint32),
        ("n_reads", C.c_uint32),
        ("read_start", C.POINTER(C.c_uint32)),
        ("read_end", C.POINTER(C.c_uint32)),
        ("row_off", C.POINTER(C.c_uint64)),
        ("alleles_2bit", C.POINTER(C.c_uint8)),
        ("quals", C.POINTER(C.c_uint8)),
        ("var_flags", C.POINTER(C.c_uint8)),
    ]


class AstarParams(C.Structure):
    _fields_ = [
        ("min_queue_size", C.c_uint64),
        ("queue_increment", C.c_uint64),
        ("max_segment_size", C.c_uint64),
        ("block_index", C.c_uint64),
    ]


class PhaseStats(C.Structure):
    _fields_ = [
        ("pruned_solutions", C.c_uint64),
        ("estimated_cost", C.c_uint64),
        ("actual_cost", C.c_uint64),
        ("phased_variants", C.c_uint64),
        ("phased_snvs", C.c_uint64),
        ("homozygous_variants", C.c_uint64),
        ("skipped_variants", C.c_uint64),
    ]

    def as_tuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


class WorkCounters(C.Structure):
    _fields_ = [
        ("sub_pops", C.c_uint64),
        ("main_pops", C.c_uint64),
        ("evals", C.c_uint64),
        ("cells", C.c_uint64),
        ("nodes_created", C.c_uint64),
        ("reserved", C.c_uint64 * 3),
    ]

    def as_tuple(self):
        return (self.sub_pops, self.main_pops, self.evals, self.cells, self.nodes_created)


class SynthSpec(C.Structure):
    _fields_ = [
        ("n_variants", C.c_uint32),
        ("coverage", C.c_uint32),
        ("span", C.c_uint32),
        ("reserved", C.c_uint32),
        ("error_rate", C.c_double),
        ("ambig_rate", C.c_double),
        ("seed", C.c_uint64),
    ]


class WfaVariant(C.Structure):
    _fields_ = [
        ("position", C.c_int64),
        ("ref_len", C.c_uint32),
        ("flags", C.c_uint32),
        ("allele0", C.POINTER(C.c_uint8)),
        ("allele0_len", C.c_uint32),
        ("allele1_len", C.c_uint32),
        ("allele1", C.POINTER(C.c_uint8)),
    ]


class WfaJob(C.Structure):
    _fields_ = [
        ("reference", C.POINTER(C.c_uint8)),
        ("ref_base", C.c_uint64),
        ("ref_start", C.c_uint64),
        ("ref_end", C.c_uint64),
        ("hets", C.POINTER(WfaVariant)),
        ("n_hets", C.c_uint32),
        ("n_homs", C.c_uint32),
        ("homs", C.POINTER(WfaVariant)),
        ("read", C.POINTER(C.c_uint8)),
        ("read_len", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class WfaResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_nodes", C.c_uint32), ("score", C.c_uint64)]


class GraphNode(C.Structure):
    _fields_ = [("seq", C.POINTER(C.c_uint8)), ("seq_len", C.c_uint32), ("n_parents", C.c_uint32), ("parents", C.POINTER(C.c_uint32))]


class GraphJob(C.Structure):
    _fields_ = [("nodes", C.POINTER(GraphNode)), ("n_nodes", C.c_uint32), ("read_len", C.c_uint32), ("read", C.POINTER(C.c_uint8))]


class GraphResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("n_traversed", C.c_uint32), ("score", C.c_uint64)]


class EdPair(C.Structure):
    _fields_ = [
        ("a", C.POINTER(C.c_uint8)),
        ("b", C.POINTER(C.c_uint8)),
        ("a_len", C.c_uint32),
        ("b_len", C.c_uint32),
    ]


class LocalVariant(C.Structure):
    _fields_ = [
        ("position", C.c_int64),
        ("ref_len", C.c_uint32),
        ("variant_type", C.c_uint32),
        ("prefix_len", C.c_uint32),
        ("postfix_len", C.c_uint32),
        ("allele0", C.POINTER(C.c_uint8)),
        ("allele1", C.POINTER(C.c_uint8)),
        ("allele0_len", C.c_uint32),
        ("allele1_len", C.c_uint32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class LocalRead(C.Structure):
    _fields_ = [
        ("pos", C.c_int64),
        ("cigar", C.POINTER(C.c_uint32)),
        ("n_cigar", C.c_uint32),
        ("seq_len", C.c_uint32),
        ("seq", C.POINTER(C.c_uint8)),
        ("qual", C.POINTER(C.c_uint8)),
        ("seq_format", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


SEQ_ASCII, SEQ_BAM4 = 0, 1
N_VARIANT_TYPES = 11


class ReadStats(C.Structure):
    _fields_ = [
        ("skipped_reads", C.c_uint64),
        ("num_alleles", C.c_uint64),
        ("exact_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("inexact_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("failed_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("allele0_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("allele1_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("local_aligned", C.c_uint64),
    ]

    def as_tuple(self):
        return (self.skipped_reads, self.num_alleles, tuple(self.exact_matches), tuple(self.inexact_matches),
                tuple(self.failed_matches), tuple(self.allele0_matches), tuple(self.allele1_matches), self.local_aligned)


class BlockRecord(C.Structure):
    _fields_ = [
        ("min_position", C.c_int64),
        ("max_position", C.c_int64),
        ("read_align", C.POINTER(C.c_uint8)),
        ("read_len", C.c_uint32),
        ("qname_id", C.c_uint32),
        ("local", C.POINTER(LocalRead)),
        ("read_offset", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class BlockInput(C.Structure):
    _fields_ = [
        ("block_index", C.c_uint64),
        ("reference", C.POINTER(C.c_uint8)),
        ("ref_base", C.c_uint64),
        ("n_hets", C.c_uint32),
        ("n_homs", C.c_uint32),
        ("n_records", C.c_uint32),
        ("n_qnames", C.c_uint32),
        ("hets", C.POINTER(WfaVariant)),
        ("het_types", C.POINTER(C.c_uint8)),
        ("local_hets", C.POINTER(LocalVariant)),
        ("homs", C.POINTER(WfaVariant)),
        ("records", C.POINTER(BlockRecord)),
        ("seq_format", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class BlockParams(C.Structure):
    _fields_ = [
        ("astar", AstarParams),
        ("wfa_prune_distance", C.c_uint64),
        ("max_edit_distance", C.c_uint64),
        ("global_failure_ratio", C.c_double),
        ("global_failure_minimum", C.c_uint64),
        ("min_matched_alleles", C.c_uint64),
        ("global_realignment", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class BlockOutput(C.Structure):
    _fields_ = [
        ("h1", C.POINTER(C.c_uint8)),
        ("h2", C.POINTER(C.c_uint8)),
        ("stats", PhaseStats),
        ("span_counts", C.POINTER(C.c_uint64)),
        ("n_segments", C.c_uint32),
        ("n_solver", C.c_uint32),
        ("seg_qname", C.POINTER(C.c_uint32)),
        ("seg_start", C.POINTER(C.c_uint32)),
        ("seg_end", C.POINTER(C.c_uint32)),
        ("seg_solver", C.POINTER(C.c_uint8)),
        ("seg_haplotag", C.POINTER(C.c_uint8)),
        ("seg_first_het", C.POINTER(C.c_uint32)),
        ("seg_row_off", C.POINTER(C.c_uint64)),
        ("seg_alleles", C.POINTER(C.c_uint8)),
        ("seg_quals", C.POINTER(C.c_uint8)),
        ("seg_cell_cap", C.c_uint64),
        ("num_reads", C.c_uint64),
        ("skipped_reads", C.c_uint64),
        ("global_aligned", C.c_uint64),
        ("local_aligned", C.c_uint64),
        ("edit_distances", C.POINTER(C.c_uint64)),
        ("n_edit_distances", C.c_uint64),
        ("status", C.c_int32),
        ("reserved", C.c_uint32),
        # the rest of the loader's ReadStats (writers/phase_stats.rs:12-33), summed over every record before the collapse
        ("num_alleles", C.c_uint64),
        ("exact_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("inexact_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("failed_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("allele0_matches", C.c_uint64 * N_VARIANT_TYPES),
        ("allele1_matches", C.c_uint64 * N_VARIANT_TYPES),
    ]

    def read_stats(self):
        """(num_alleles, exact, inexact, failed, allele0, allele1) as plain tuples"""
        return (int(self.num_alleles), tuple(self.exact_matches), tuple(self.inexact_matches), tuple(self.failed_matches),
                tuple(self.allele0_matches), tuple(self.allele1_matches))


class SynthReadsSpec(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64),
        ("total_hets", C.c_uint32),
        ("max_block_hets", C.c_uint32),
        ("coverage", C.c_double), ("read_mean", C.c_double), ("read_sd", C.c_double), ("het_spacing", C.c_double), ("hom_ratio", C.c_double),
        ("frac_snv", C.c_double), ("frac_indel", C.c_double), ("frac_sv", C.c_double), ("frac_multiallelic", C.c_double),
        ("edit_noise", C.c_double), ("noisy_fraction", C.c_double), ("noisy_noise", C.c_double), ("supplementary_fraction", C.c_double),
        ("seq_format", C.c_uint32),
        ("threads", C.c_uint32),
        ("hifi_sigma", C.c_double), ("homopolymer_share", C.c_double),
        ("allele_switch", C.c_double),
    ]


# Every symbol include/hiphase_gpu.h declares; tests check the library exports all of them.
EXPORTS = [
    "hp_astar_solve",
    "hp_astar_solve_batch",
    "hp_batch_create",
    "hp_batch_solve",
    "hp_batch_results",
    "hp_batch_postprocess",
    "hp_batch_destroy",
    "hp_wfa_assign_batch",
    "hp_wfa_align_graphs",
    "hp_edit_distance_batch",
    "hp_local_realign_batch",
    "hp_solve_blocks",
    "hp_blockset_create",
    "hp_blockset_solve",
    "hp_blockset_work",
    "hp_blockset_destroy",
    "hp_blockstream_create",
    "hp_blockstream_submit",
    "hp_blockstream_wait",
    "hp_blockstream_destroy",
    "hp_blockstream_devices",
    "hp_synth_reads_hifi",
    "hp_synth_reads_deep60",
    "hp_block_submit",
    "hp_block_wait",
    "hp_device_count",
    "hp_default_device",
    "hp_last_error",
    "hp_version",
    "hp_set_coalescing",
    "hp_last_kernel_ms",
    "hp_runtime_wait_mode",
    "hp_trim_device_cache",
    "hp_host_alloc",
    "hp_host_free",
    "hp_host_in_place_bytes",
    "hp_wfa_routed_records",
    "hp_abi_layout",
    "hp_abi_sizeof",
    "hp_abi_offsetof",
    "hp_hpbk_append",
    "hp_synth_block_size",
    "hp_synth_block",
    "hp_synth_reads_defaults",
    "hp_synth_reads_create",
    "hp_synth_reads_inputs",
    "hp_synth_reads_info",
    "hp_synth_reads_truth",
    "hp_synth_reads_relocate",
    "hp_synth_reads_destroy",
    "hp_outputs_create",
    "hp_outputs_array",
    "hp_outputs_destroy",
    "hp_outputs_poison",
    "hp_block_output_equal",
    "hp_hpbr_append",
    "hp_hpbr_open",
    "hp_hpbr_inputs",
    "hp_hpbr_params",
    "hp_hpbr_expected",
    "hp_hpbr_close",
    "hp_hpbr_last_error",
]


def declare_common(dll):
    """argtypes/restype for the symbols shared by libhiphase_gpu.so and (synth only) liboracle.so."""
    dll.hp_synth_block_size.restype = C.c_uint32
    dll.hp_synth_block_size.argtypes = [C.POINTER(SynthSpec), C.POINTER(C.c_uint64)]
    dll.hp_synth_block.restype = C.c_int
    dll.hp_synth_block.argtypes = [C.POINTER(SynthSpec)] + [C.c_void_p] * 7
    dll.hp_synth_reads_defaults.restype = None
    dll.hp_synth_reads_defaults.argtypes = [C.POINTER(SynthReadsSpec)]
    dll.hp_synth_reads_hifi.argtypes = [C.POINTER(SynthReadsSpec)]
    dll.hp_synth_reads_hifi.restype = None
    dll.hp_synth_reads_deep60.argtypes = [C.POINTER(SynthReadsSpec)]
    dll.hp_synth_reads_deep60.restype = None
    dll.hp_synth_reads_create.restype = C.c_void_p
    dll.hp_synth_reads_create.argtypes = [C.POINTER(SynthReadsSpec), C.POINTER(C.c_int)]
    dll.hp_synth_reads_inputs.restype = C.POINTER(BlockInput)
    dll.hp_synth_reads_inputs.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    dll.hp_synth_reads_info.restype = None
    dll.hp_synth_reads_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    dll.hp_synth_reads_truth.restype = C.POINTER(C.c_uint8)
    dll.hp_synth_reads_truth.argtypes = [C.c_void_p, C.c_size_t]
    dll.hp_synth_reads_relocate.restype = C.c_int
    dll.hp_synth_reads_relocate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    dll.hp_synth_reads_destroy.restype = None
    dll.hp_synth_reads_destroy.argtypes = [C.c_void_p]
    dll.hp_outputs_create.restype = C.c_void_p
    dll.hp_outputs_create.argtypes = [C.POINTER(BlockInput), C.c_size_t]
    dll.hp_outputs_array.restype = C.POINTER(BlockOutput)
    dll.hp_outputs_array.argtypes = [C.c_void_p]
    dll.hp_outputs_destroy.restype = None
    dll.hp_outputs_destroy.argtypes = [C.c_void_p]
    dll.hp_outputs_poison.restype = None
    dll.hp_outputs_poison.argtypes = [C.c_void_p, C.c_uint8]
    dll.hp_hpbr_append.restype = C.c_int
    dll.hp_hpbr_append.argtypes = [C.c_char_p, C.POINTER(BlockInput), C.POINTER(BlockParams), C.POINTER(BlockOutput)]
    dll.hp_hpbr_open.restype = C.c_void_p
    dll.hp_hpbr_open.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    dll.hp_hpbr_inputs.restype = C.POINTER(BlockInput)
    dll.hp_hpbr_inputs.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    dll.hp_hpbr_params.restype = C.POINTER(BlockParams)
    dll.hp_hpbr_params.argtypes = [C.c_void_p]
    dll.hp_hpbr_expected.restype = C.POINTER(BlockOutput)
    dll.hp_hpbr_expected.argtypes = [C.c_void_p]
    dll.hp_hpbr_close.restype = None
    dll.hp_hpbr_close.argtypes = [C.c_void_p]
    dll.hp_hpbr_last_error.restype = C.c_char_p
    dll.hp_block_output_equal.restype = C.c_int
    dll.hp_block_output_equal.argtypes = [C.POINTER(BlockInput), C.POINTER(BlockOutput), C.POINTER(BlockOutput)]


_lib = None


def lib():
    """Load libhiphase_gpu.so (once). Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HpError(-1, f"{LIB_PATH} not found — run `python -c 'import __graft_entry__ as g; g.build()'`; "
                          "there is no CPU fallback")
    dll = C.CDLL(LIB_PATH)
    declare_common(dll)
    dll.hp_astar_solve.restype = C.c_int
    dll.hp_astar_solve.argtypes = [C.POINTER(BlockView), C.POINTER(AstarParams), C.c_void_p, C.c_void_p,
                                   C.POINTER(PhaseStats)]
    dll.hp_astar_solve_batch.restype = C.c_int
    dll.hp_astar_solve_batch.argtypes = [C.c_size_t, C.POINTER(BlockView), C.POINTER(AstarParams),
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(PhaseStats), C.c_int]
    dll.hp_batch_create.restype = C.c_void_p
    dll.hp_batch_create.argtypes = [C.c_size_t, C.POINTER(BlockView), C.POINTER(AstarParams), C.c_int,
                                    C.POINTER(C.c_int)]
    dll.hp_batch_solve.restype = C.c_int
    dll.hp_batch_solve.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    dll.hp_batch_results.restype = C.c_int
    dll.hp_batch_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dll.hp_batch_postprocess.restype = C.c_int
    dll.hp_batch_postprocess.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dll.hp_batch_destroy.restype = None
    dll.hp_batch_destroy.argtypes = [C.c_void_p]
    dll.hp_wfa_assign_batch.restype = C.c_int
    dll.hp_wfa_assign_batch.argtypes = [C.POINTER(WfaJob), C.c_size_t, C.c_uint64, C.c_uint64,
                                        C.POINTER(WfaResult), C.POINTER(C.c_void_p), C.c_int]
    dll.hp_wfa_align_graphs.restype = C.c_int
    dll.hp_wfa_align_graphs.argtypes = [C.POINTER(GraphJob), C.c_size_t, C.c_uint64, C.c_uint64, C.POINTER(GraphResult),
                                        C.POINTER(C.c_void_p), C.c_int]
    dll.hp_edit_distance_batch.restype = C.c_int
    dll.hp_edit_distance_batch.argtypes = [C.POINTER(EdPair), C.c_size_t, C.POINTER(C.c_uint64), C.c_int]
    dll.hp_local_realign_batch.restype = C.c_int
    dll.hp_local_realign_batch.argtypes = [C.POINTER(LocalRead), C.c_size_t, C.POINTER(LocalVariant), C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.POINTER(ReadStats), C.c_int]
    dll.hp_solve_blocks.restype = C.c_int
    dll.hp_solve_blocks.argtypes = [C.c_size_t, C.POINTER(BlockInput), C.POINTER(BlockParams), C.POINTER(BlockOutput), C.c_int]
    dll.hp_blockset_create.restype = C.c_void_p
    dll.hp_blockset_create.argtypes = [C.c_size_t, C.POINTER(BlockInput), C.POINTER(BlockParams), C.c_int, C.POINTER(C.c_int)]
    dll.hp_blockset_solve.restype = C.c_int
    dll.hp_blockset_solve.argtypes = [C.c_void_p, C.POINTER(BlockOutput), C.POINTER(C.c_double)]
    dll.hp_blockset_work.restype = C.c_int
    dll.hp_blockset_work.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    dll.hp_blockset_destroy.restype = None
    dll.hp_blockset_destroy.argtypes = [C.c_void_p]
    dll.hp_blockstream_create.restype = C.c_void_p
    dll.hp_blockstream_create.argtypes = [C.POINTER(BlockParams), C.c_int, C.c_uint32, C.POINTER(C.c_int)]
    dll.hp_blockstream_submit.restype = C.c_int
    dll.hp_blockstream_submit.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(BlockInput), C.POINTER(BlockOutput), C.POINTER(C.c_uint64)]
    dll.hp_blockstream_devices.restype = C.c_int
    dll.hp_blockstream_devices.argtypes = [C.c_void_p]
    dll.hp_block_submit.restype = C.c_int
    dll.hp_block_submit.argtypes = [C.c_size_t, C.POINTER(BlockInput), C.POINTER(BlockParams), C.POINTER(BlockOutput), C.c_int, C.POINTER(C.c_uint64)]
    dll.hp_block_wait.restype = C.c_int
    dll.hp_block_wait.argtypes = [C.c_uint64]
    dll.hp_blockstream_wait.restype = C.c_int
    dll.hp_blockstream_wait.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    dll.hp_blockstream_destroy.restype = None
    dll.hp_blockstream_destroy.argtypes = [C.c_void_p]
    dll.hp_abi_layout.restype = C.c_char_p
    dll.hp_hpbk_append.restype = C.c_int
    dll.hp_hpbk_append.argtypes = [C.c_char_p, C.POINTER(BlockView), C.POINTER(AstarParams), C.c_void_p, C.c_void_p, C.POINTER(PhaseStats)]
    dll.hp_set_coalescing.restype = C.c_int
    dll.hp_set_coalescing.argtypes = [C.c_int]
    dll.hp_device_count.restype = C.c_int
    dll.hp_default_device.restype = C.c_int
    dll.hp_last_error.restype = C.c_char_p
    dll.hp_version.restype = C.c_char_p
    dll.hp_last_kernel_ms.restype = C.c_double
    dll.hp_runtime_wait_mode.restype = C.c_int
    dll.hp_trim_device_cache.restype = C.c_size_t
    dll.hp_host_alloc.restype = C.c_void_p
    dll.hp_host_alloc.argtypes = [C.c_size_t]
    dll.hp_host_in_place_bytes.restype = C.c_uint64
    if hasattr(dll, "hp_wfa_routed_records"):   # (HP_LIB may name an older build of the library)
        dll.hp_wfa_routed_records.restype = C.c_uint64
    dll.hp_host_free.restype = None
    dll.hp_host_free.argtypes = [C.c_void_p]
    _lib = dll
    return dll


def check(code):
    if code < 0:
        raise HpError(code, lib().hp_last_error().decode())
    return code
