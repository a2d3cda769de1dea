"""ctypes binding of the C ABI in include/pll.h + include/pllhip.h, and the
synthetic workload generator shared by tests/ and bench.py.

The binding is deliberately thin: it mirrors the C structs field by field
(`pll_partition_t`, `pll_operation_t`, `pll_unode_t`, `pll_utree_t`) and calls
the exported functions with plain pointers and sizes, i.e. exactly what a C
caller such as pll-modules' treeinfo.c does.  `PllLib(path)` can wrap any
shared library that exports the interface; this module itself only knows where
the PRODUCT library lives (`PRODUCT_LIB`).  The CPU oracle is loaded by tests/
and by bench.py's cpu_baseline leg, never from here.

Workloads follow SURVEY.md section 8d: splitmix64-seeded random stepwise-addition
tree (seed 42), branch lengths U(0.01, 0.2) (seed 43), iid uniform tip states
(seed 44), DNA GTR+G4 / "LG-shaped" protein / GY94-shaped codon models.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(HERE, "libpll_hip.so")

PLL_SCALE_BUFFER_NONE = -1
PLL_ATTRIB_PATTERN_TIP = 1 << 4
PLL_ATTRIB_AB_LEWIS, PLL_ATTRIB_AB_FELSENSTEIN, PLL_ATTRIB_AB_STAMATAKIS = 1 << 5, 2 << 5, 3 << 5
PLL_ATTRIB_AB_FLAG = 1 << 8
PLL_ATTRIB_RATE_SCALERS = 1 << 9
PLL_ATTRIB_SITE_REPEATS = 1 << 10
PLL_GAMMA_RATES_MEAN = 0
PLL_TREE_TRAVERSE_POSTORDER = 1
PLLHIP_SYNC_PMATRIX, PLLHIP_SYNC_CLV, PLLHIP_SYNC_SCALERS, PLLHIP_SYNC_TIPS, PLLHIP_SYNC_ALL = 1, 2, 4, 8, 15
PLLHIP_ATTRIB_HOST_MIRRORS = 1 << 30

c_double_p = C.POINTER(C.c_double)
c_uint_p = C.POINTER(C.c_uint)


class Repeats(C.Structure):
    """pll_repeats_t (include/pll.h): what pll-modules dereferences under PLL_ATTRIB_SITE_REPEATS"""
    _fields_ = [("pernode_site_id", C.POINTER(c_uint_p)), ("pernode_id_site", C.POINTER(c_uint_p)),
                ("pernode_ids", c_uint_p), ("perscale_ids", c_uint_p), ("pernode_allocated_clvs", c_uint_p),
                ("enable_repeats", C.c_void_p), ("reallocate_repeats", C.c_void_p)]


class Partition(C.Structure):
    _fields_ = [
        ("tips", C.c_uint), ("clv_buffers", C.c_uint), ("nodes", C.c_uint),
        ("states", C.c_uint), ("sites", C.c_uint), ("pattern_weight_sum", C.c_uint),
        ("rate_matrices", C.c_uint), ("prob_matrices", C.c_uint), ("rate_cats", C.c_uint),
        ("scale_buffers", C.c_uint), ("attributes", C.c_uint),
        ("alignment", C.c_size_t), ("states_padded", C.c_uint),
        ("clv", C.POINTER(c_double_p)), ("pmatrix", C.POINTER(c_double_p)),
        ("rates", c_double_p), ("rate_weights", c_double_p),
        ("subst_params", C.POINTER(c_double_p)),
        ("scale_buffer", C.POINTER(c_uint_p)),
        ("frequencies", C.POINTER(c_double_p)),
        ("prop_invar", c_double_p), ("invariant", C.POINTER(C.c_int)),
        ("pattern_weights", c_uint_p),
        ("eigen_decomp_valid", C.POINTER(C.c_int)),
        ("eigenvecs", C.POINTER(c_double_p)), ("inv_eigenvecs", C.POINTER(c_double_p)),
        ("eigenvals", C.POINTER(c_double_p)),
        ("maxstates", C.c_uint), ("tipchars", C.POINTER(C.POINTER(C.c_ubyte))),
        ("charmap", C.POINTER(C.c_ubyte)), ("ttlookup", c_double_p),
        ("tipmap", C.POINTER(C.c_ulonglong)),
        ("asc_bias_alloc", C.c_int), ("asc_additional_sites", C.c_int),
        ("repeats", C.POINTER(Repeats)), ("engine", C.c_void_p),
    ]


class Operation(C.Structure):
    _fields_ = [
        ("parent_clv_index", C.c_uint), ("parent_scaler_index", C.c_int),
        ("child1_clv_index", C.c_uint), ("child1_matrix_index", C.c_uint),
        ("child1_scaler_index", C.c_int),
        ("child2_clv_index", C.c_uint), ("child2_matrix_index", C.c_uint),
        ("child2_scaler_index", C.c_int),
    ]


class UNode(C.Structure):
    pass


UNode._fields_ = [
    ("label", C.c_char_p), ("length", C.c_double), ("node_index", C.c_uint),
    ("clv_index", C.c_uint), ("scaler_index", C.c_int), ("pmatrix_index", C.c_uint),
    ("next", C.POINTER(UNode)), ("back", C.POINTER(UNode)), ("data", C.c_void_p),
]


class UTree(C.Structure):
    _fields_ = [
        ("tip_count", C.c_uint), ("inner_count", C.c_uint), ("edge_count", C.c_uint),
        ("binary", C.c_int), ("nodes", C.POINTER(C.POINTER(UNode))),
        ("vroot", C.POINTER(UNode)),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in (
        "partial_ops", "partial_launches", "site_updates", "pmatrix_updates", "pmatrix_launches",
        "lnl_calls", "sumtable_calls", "derivative_calls", "derivative_points", "model_uploads")]


class RepeatStats(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("cherries", "classes", "sites", "expansions")]


class TransientStats(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("skipped", "materialized", "discarded")]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_ulonglong), ("ops", C.c_ulonglong),
                ("kernel_ms", C.c_double), ("algorithmic_bytes", C.c_double),
                ("algorithmic_flops", C.c_double), ("minimum_bytes", C.c_double)]


TRAVERSE_CB = C.CFUNCTYPE(C.c_int, C.POINTER(UNode))
REDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, c_double_p, C.c_size_t, C.c_int)

# every function the headers declare with PLL_EXPORT, for the symbol test
PLL_H_FUNCTIONS = """pll_partition_create pll_partition_destroy pll_set_tip_states
pll_set_tip_clv pll_set_pattern_weights pll_set_asc_bias_type pll_set_asc_state_weights
pll_set_subst_params pll_set_frequencies pll_set_category_rates pll_set_category_weights
pll_update_eigen pll_count_invariant_sites pll_update_invariant_sites
pll_update_invariant_sites_proportion pll_compute_gamma_cats pll_aligned_alloc
pll_aligned_free pll_get_sites_number pll_get_clv_size pll_update_prob_matrices
pll_update_partials pll_compute_root_loglikelihood pll_compute_edge_loglikelihood
pll_update_sumtable pll_compute_likelihood_derivatives pll_compute_node_ancestral
pll_show_pmatrix pll_show_clv pll_utree_traverse pll_utree_create_operations
pll_utree_wraptree pll_utree_wraptree_multi pll_utree_destroy pll_utree_graph_clone
pll_utree_graph_destroy pll_utree_clone pll_utree_reset_template_indices
pll_utree_check_integrity pll_utree_every pll_utree_parse_newick
pll_utree_parse_newick_unroot pll_utree_parse_newick_string
pll_utree_parse_newick_string_unroot pll_utree_export_newick pll_utree_show_ascii
pll_random_create pll_random_getint pll_random_destroy""".split()

PLLHIP_EVAL_H_FUNCTIONS = """pllhip_eval_create pllhip_eval_destroy pllhip_eval_set_partition
pllhip_eval_set_parallel_context pllhip_eval_set_root pllhip_eval_root pllhip_eval_invalidate_all
pllhip_eval_invalidate_pmatrix pllhip_eval_invalidate_clv pllhip_eval_loglh
pllhip_eval_set_branch_length pllhip_eval_optimize_branches pllhip_eval_ops
pllhip_eval_pmatrix_updates pllhip_eval_derivative_calls pllhip_eval_spr_round
pllhip_eval_set_fused pllhip_eval_newton_iterations pllhip_eval_set_brlen_linkage
pllhip_eval_set_brlen_scaler pllhip_eval_get_brlen_scaler pllhip_eval_get_partition_branch_length
pllhip_eval_set_partition_branch_length pllhip_eval_set_transient""".split()

PLLHIP_H_FUNCTIONS = """pllhip_device_count pllhip_set_device pllhip_get_device
pllhip_device_arch pllhip_eigen_decompose pllhip_sync_to_host pllhip_sync_to_device pllhip_get_clv
pllhip_get_scaler pllhip_get_sumtable pllhip_set_clv pllhip_set_scaler pllhip_synchronize
pllhip_stream pllhip_get_counters pllhip_reset_counters pllhip_partials_kernel_name
pllhip_comm_get_unique_id pllhip_comm_create pllhip_comm_destroy pllhip_reduce_cb
pllhip_profile_partials pllhip_profile_read pllhip_comm_rank pllhip_comm_size
pllhip_compute_likelihood_derivatives_multi pllhip_free_trial_lengths pllhip_set_sharding
pllhip_shard_count pllhip_results_create pllhip_results_destroy
pllhip_results_edge_loglikelihood pllhip_results_derivatives pllhip_results_fetch
pllhip_eval_attach_comm pllhip_update_partials_batch pllhip_results_poison pllhip_newton_branch pllhip_repeat_stats
pllhip_set_transient pllhip_discard_transient pllhip_transient_stats pllhip_newton_branch_multi""".split()


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class PllLib:
    """One loaded implementation of include/pll.h."""

    def __init__(self, path=PRODUCT_LIB):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        self.lib = L = C.CDLL(path)
        self.is_product = hasattr(L, "pllhip_device_count")
        pp = C.POINTER(Partition)
        L.pll_partition_create.restype = pp
        L.pll_partition_create.argtypes = [C.c_uint] * 9
        L.pll_partition_destroy.argtypes = [pp]
        L.pll_get_sites_number.argtypes = [pp, C.c_uint]
        L.pll_get_sites_number.restype = C.c_uint
        L.pll_get_clv_size.argtypes = [pp, C.c_uint]
        L.pll_get_clv_size.restype = C.c_uint
        L.pll_set_tip_states.argtypes = [pp, C.c_uint, C.POINTER(C.c_ulonglong), C.c_char_p]
        L.pll_set_tip_clv.argtypes = [pp, C.c_uint, c_double_p, C.c_int]
        L.pll_set_pattern_weights.argtypes = [pp, c_uint_p]
        L.pll_set_asc_bias_type.argtypes = [pp, C.c_int]
        L.pll_set_asc_state_weights.argtypes = [pp, c_uint_p]
        L.pll_set_asc_state_weights.restype = None
        L.pll_set_subst_params.argtypes = [pp, C.c_uint, c_double_p]
        L.pll_set_frequencies.argtypes = [pp, C.c_uint, c_double_p]
        L.pll_set_category_rates.argtypes = [pp, c_double_p]
        L.pll_set_category_weights.argtypes = [pp, c_double_p]
        L.pll_update_eigen.argtypes = [pp, C.c_uint]
        L.pll_update_invariant_sites.argtypes = [pp]
        L.pll_update_invariant_sites_proportion.argtypes = [pp, C.c_uint, C.c_double]
        L.pll_count_invariant_sites.argtypes = [pp, c_uint_p]
        L.pll_count_invariant_sites.restype = C.c_uint
        L.pll_compute_gamma_cats.argtypes = [C.c_double, C.c_uint, c_double_p, C.c_int]
        L.pll_update_prob_matrices.argtypes = [pp, c_uint_p, c_uint_p, c_double_p, C.c_uint]
        L.pll_update_partials.argtypes = [pp, C.POINTER(Operation), C.c_uint]
        L.pll_update_partials.restype = None
        L.pll_compute_edge_loglikelihood.restype = C.c_double
        L.pll_compute_edge_loglikelihood.argtypes = [pp, C.c_uint, C.c_int, C.c_uint, C.c_int,
                                                     C.c_uint, c_uint_p, c_double_p]
        L.pll_compute_root_loglikelihood.restype = C.c_double
        L.pll_compute_root_loglikelihood.argtypes = [pp, C.c_uint, C.c_int, c_uint_p, c_double_p]
        L.pll_update_sumtable.argtypes = [pp, C.c_uint, C.c_uint, C.c_int, C.c_int, c_uint_p, c_double_p]
        L.pll_compute_likelihood_derivatives.argtypes = [pp, C.c_int, C.c_int, C.c_double, c_uint_p,
                                                         c_double_p, c_double_p, c_double_p]
        L.pll_compute_node_ancestral.argtypes = [pp, C.c_uint, C.c_int, C.c_uint, C.c_int, C.c_uint,
                                                 c_uint_p, c_double_p]
        L.pll_aligned_alloc.restype = C.c_void_p
        L.pll_aligned_alloc.argtypes = [C.c_size_t, C.c_size_t]
        L.pll_aligned_free.argtypes = [C.c_void_p]
        up, tp = C.POINTER(UNode), C.POINTER(UTree)
        L.pll_utree_parse_newick_string.restype = tp
        L.pll_utree_parse_newick_string.argtypes = [C.c_char_p]
        L.pll_utree_parse_newick_string_unroot.restype = tp
        L.pll_utree_parse_newick_string_unroot.argtypes = [C.c_char_p]
        L.pll_utree_destroy.argtypes = [tp, C.c_void_p]
        L.pll_utree_clone.restype = tp
        L.pll_utree_clone.argtypes = [tp]
        L.pll_utree_check_integrity.argtypes = [tp]
        L.pll_utree_traverse.argtypes = [up, C.c_int, TRAVERSE_CB, C.POINTER(up), c_uint_p]
        L.pll_utree_create_operations.restype = None
        L.pll_utree_create_operations.argtypes = [C.POINTER(up), C.c_uint, c_double_p, c_uint_p,
                                                  C.POINTER(Operation), c_uint_p, c_uint_p]
        L.pll_utree_export_newick.restype = C.c_void_p
        L.pll_utree_export_newick.argtypes = [up, C.c_void_p]
        if hasattr(L, "pllhip_eval_create"):
            L.pllhip_eval_create.restype = C.c_void_p
            L.pllhip_eval_create.argtypes = [tp, C.c_uint, C.c_uint]
            L.pllhip_eval_destroy.argtypes = [C.c_void_p]
            L.pllhip_eval_set_partition.argtypes = [C.c_void_p, C.c_uint, pp, c_uint_p]
            L.pllhip_eval_set_parallel_context.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            L.pllhip_eval_set_root.argtypes = [C.c_void_p, up]
            L.pllhip_eval_root.restype = up
            L.pllhip_eval_root.argtypes = [C.c_void_p]
            L.pllhip_eval_invalidate_all.argtypes = [C.c_void_p]
            L.pllhip_eval_invalidate_pmatrix.argtypes = [C.c_void_p, up]
            L.pllhip_eval_invalidate_clv.argtypes = [C.c_void_p, up]
            L.pllhip_eval_loglh.restype = C.c_double
            L.pllhip_eval_loglh.argtypes = [C.c_void_p, C.c_int]
            L.pllhip_eval_set_branch_length.argtypes = [C.c_void_p, up, C.c_double]
            L.pllhip_eval_set_transient.argtypes = [C.c_void_p, C.c_int]
            L.pllhip_eval_set_transient.restype = None
            L.pllhip_eval_optimize_branches.restype = C.c_double
            L.pllhip_eval_optimize_branches.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double,
                                                        C.c_int, C.c_int]
            L.pllhip_eval_set_brlen_linkage.argtypes = [C.c_void_p, C.c_int]
            L.pllhip_eval_set_brlen_scaler.argtypes = [C.c_void_p, C.c_uint, C.c_double]
            L.pllhip_eval_get_brlen_scaler.restype = C.c_double
            L.pllhip_eval_get_brlen_scaler.argtypes = [C.c_void_p, C.c_uint]
            L.pllhip_eval_get_partition_branch_length.restype = C.c_double
            L.pllhip_eval_get_partition_branch_length.argtypes = [C.c_void_p, C.c_uint, up]
            L.pllhip_eval_set_partition_branch_length.argtypes = [C.c_void_p, C.c_uint, up, C.c_double]
            L.pllhip_eval_spr_round.restype = C.c_double
            L.pllhip_eval_spr_round.argtypes = [C.c_void_p, C.POINTER(SprParams), C.POINTER(SprCutoff),
                                                C.POINTER(SprStats)]
            for fn in ("pllhip_eval_ops", "pllhip_eval_pmatrix_updates", "pllhip_eval_derivative_calls",
                       "pllhip_eval_newton_iterations"):
                getattr(L, fn).restype = C.c_ulong
                getattr(L, fn).argtypes = [C.c_void_p]
        if self.is_product:
            L.pllhip_device_count.restype = C.c_int
            L.pllhip_set_device.argtypes = [C.c_int]
            L.pllhip_device_arch.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
            L.pllhip_eigen_decompose.argtypes = [C.c_uint, C.c_uint, c_double_p, c_double_p,
                                                 c_double_p, c_double_p, c_double_p]
            L.pllhip_sync_to_host.argtypes = [pp, C.c_uint]
            L.pllhip_sync_to_device.argtypes = [pp, C.c_uint]
            L.pllhip_get_clv.argtypes = [pp, C.c_uint, c_double_p]
            L.pllhip_set_clv.argtypes = [pp, C.c_uint, c_double_p]
            L.pllhip_get_scaler.argtypes = [pp, C.c_uint, c_uint_p]
            L.pllhip_set_scaler.argtypes = [pp, C.c_uint, c_uint_p]
            L.pllhip_get_sumtable.argtypes = [pp, c_double_p, c_double_p]
            L.pllhip_synchronize.argtypes = [pp]
            L.pllhip_stream.restype = C.c_void_p
            L.pllhip_stream.argtypes = [pp]
            L.pllhip_get_counters.argtypes = [pp, C.POINTER(Counters)]
            L.pllhip_reset_counters.argtypes = [pp]
            L.pllhip_partials_kernel_name.restype = C.c_char_p
            L.pllhip_partials_kernel_name.argtypes = [pp]
            L.pllhip_profile_partials.argtypes = [pp, C.c_int]
            L.pllhip_profile_read.argtypes = [pp, C.POINTER(Profile)]
            L.pllhip_repeat_stats.argtypes = [pp, C.POINTER(RepeatStats)]
            L.pllhip_set_transient.argtypes = [pp, C.c_int]
            L.pllhip_discard_transient.argtypes = [pp]
            L.pllhip_transient_stats.argtypes = [pp, C.POINTER(TransientStats)]
            L.pllhip_comm_get_unique_id.argtypes = [C.c_char_p]
            L.pllhip_comm_create.restype = C.c_void_p
            L.pllhip_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
            L.pllhip_comm_destroy.argtypes = [C.c_void_p]
            L.pllhip_reduce_cb.restype = None
            L.pllhip_reduce_cb.argtypes = [C.c_void_p, c_double_p, C.c_size_t, C.c_int]
            L.pllhip_set_sharding.argtypes = [C.c_uint, C.POINTER(C.c_int)]
            L.pllhip_shard_count.argtypes = [pp]
            L.pllhip_shard_count.restype = C.c_uint
            L.pllhip_comm_rank.argtypes = [C.c_void_p]
            L.pllhip_comm_size.argtypes = [C.c_void_p]
            L.pllhip_eval_attach_comm.argtypes = [C.c_void_p, C.c_void_p]
            L.pllhip_eval_set_fused.argtypes = [C.c_void_p, C.c_void_p]
            L.pllhip_eval_set_fused.restype = None
            L.pllhip_results_create.restype = C.c_void_p
            L.pllhip_results_create.argtypes = [C.c_void_p, C.c_uint]
            L.pllhip_results_destroy.argtypes = [C.c_void_p]
            L.pllhip_results_edge_loglikelihood.argtypes = [C.c_void_p, C.c_uint, pp, C.c_uint, C.c_int,
                                                            C.c_uint, C.c_int, C.c_uint, c_uint_p]
            L.pllhip_results_derivatives.argtypes = [C.c_void_p, C.c_uint, pp, C.c_int, C.c_int, c_double_p,
                                                     C.c_uint, c_uint_p, c_double_p]
            L.pllhip_results_fetch.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int, c_double_p]
            L.pllhip_results_poison.argtypes = [C.c_void_p]
            L.pllhip_results_poison.restype = None
        if hasattr(L, "pllhip_newton_branch"):
            L.pllhip_newton_branch.argtypes = [pp, C.c_int, C.c_int, c_uint_p, c_double_p, C.c_double, C.c_double,
                                               C.c_double, C.c_double, C.c_uint, c_double_p, c_uint_p, c_double_p]
        if hasattr(L, "pllhip_newton_branch_multi"):
            L.pllhip_newton_branch_multi.argtypes = [C.POINTER(pp), C.c_uint, C.c_int, C.c_int, C.POINTER(c_uint_p),
                                                     C.POINTER(c_double_p), c_double_p, C.c_double, C.c_double, C.c_double,
                                                     C.c_double, C.c_uint, c_double_p, c_uint_p, c_double_p]
        if hasattr(L, "pllhip_update_partials_batch"):
            L.pllhip_update_partials_batch.argtypes = [C.POINTER(pp), C.c_uint, C.POINTER(Operation), C.c_uint]
        if hasattr(L, "pllhip_compute_likelihood_derivatives_multi"):
            L.pllhip_compute_likelihood_derivatives_multi.argtypes = [pp, C.c_int, C.c_int, c_double_p, C.c_uint,
                                                                      c_uint_p, c_double_p, c_double_p, c_double_p]

    # --- error state ------------------------------------------------------
    @property
    def errno(self):
        return C.c_int.in_dll(self.lib, "pll_errno").value

    @errno.setter
    def errno(self, v):
        C.c_int.in_dll(self.lib, "pll_errno").value = v

    @property
    def errmsg(self):
        return (C.c_char * 200).in_dll(self.lib, "pll_errmsg").value.decode(errors="replace")

    def gamma_cats(self, alpha, k, mode=PLL_GAMMA_RATES_MEAN):
        out = np.zeros(k)
        if not self.lib.pll_compute_gamma_cats(alpha, k, out.ctypes.data_as(c_double_p), mode):
            raise RuntimeError(self.errmsg)
        return out


class Instance:
    """A partition plus the index bookkeeping of one (tree, model, alignment)."""

    def __init__(self, lib, tips, states, sites, rate_cats, attributes=0, scalers=True,
                 rate_matrices=1, prob_matrices=None, clv_buffers=None):
        self.lib, self.L = lib, lib.lib
        self.tips, self.S, self.N, self.R = tips, states, sites, rate_cats
        inner = tips - 2 if clv_buffers is None else clv_buffers
        nmat = 2 * tips - 3 if prob_matrices is None else prob_matrices
        self.nscalers = inner if scalers else 0
        self.p = self.L.pll_partition_create(tips, inner, states, sites, rate_matrices, nmat,
                                             rate_cats, self.nscalers, attributes)
        if not self.p:
            raise RuntimeError(f"pll_partition_create failed: [{lib.errno}] {lib.errmsg}")
        self.Sp = self.p.contents.states_padded
        self.rate_scalers = bool(attributes & PLL_ATTRIB_RATE_SCALERS)
        # with ascertainment-bias correction every per-site array carries `states` extra patterns
        self.Nalloc = sites + (states if attributes & (PLL_ATTRIB_AB_FLAG | (7 << 5)) else 0)
        self.params = _u32(np.zeros(rate_cats))
        self._keep = []

    def close(self):
        if self.p:
            self.L.pll_partition_destroy(self.p)
            self.p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def params_p(self):
        return self.params.ctypes.data_as(c_uint_p)

    def set_params_indices(self, indices):
        """per-rate-category parameter-set indices (params_indices / freqs_indices of every kernel
        call; src/tree/treeinfo.c:288-306): category r uses rate matrix indices[r]"""
        idx = _u32(indices)
        assert len(idx) == self.R
        self.params = idx

    # --- model ------------------------------------------------------------
    def set_model(self, subst, freqs, rates, weights=None, idx=0):
        s, f, r = _f64(subst), _f64(freqs), _f64(rates)
        self.L.pll_set_subst_params(self.p, idx, s.ctypes.data_as(c_double_p))
        self.L.pll_set_frequencies(self.p, idx, f.ctypes.data_as(c_double_p))
        self.L.pll_set_category_rates(self.p, r.ctypes.data_as(c_double_p))
        if weights is not None:
            w = _f64(weights)
            self.L.pll_set_category_weights(self.p, w.ctypes.data_as(c_double_p))

    def set_tip_states(self, tip, charmap, seq_bytes):
        m = (C.c_ulonglong * 256)(*[int(x) for x in charmap])
        if not self.L.pll_set_tip_states(self.p, tip, m, bytes(seq_bytes)):
            raise RuntimeError(f"pll_set_tip_states: {self.lib.errmsg}")

    def set_tip_clv(self, tip, clv, padding=0):
        a = _f64(clv)
        if not self.L.pll_set_tip_clv(self.p, tip, a.ctypes.data_as(c_double_p), padding):
            raise RuntimeError(f"pll_set_tip_clv: {self.lib.errmsg}")

    def set_pattern_weights(self, w):
        a = _u32(w)
        self.L.pll_set_pattern_weights(self.p, a.ctypes.data_as(c_uint_p))

    def set_asc(self, asc_type, state_weights=None):
        if not self.L.pll_set_asc_bias_type(self.p, asc_type):
            raise RuntimeError(self.lib.errmsg)
        if state_weights is not None:
            w = _u32(state_weights)
            self.L.pll_set_asc_state_weights(self.p, w.ctypes.data_as(c_uint_p))

    def set_pinv(self, pinv, idx=0):
        if not self.L.pll_update_invariant_sites_proportion(self.p, idx, pinv):
            raise RuntimeError(self.lib.errmsg)

    # --- hot path ---------------------------------------------------------
    def update_pmatrices(self, matrix_indices, brlens, one_by_one=False):
        mi, bl = _u32(matrix_indices), _f64(brlens)
        if one_by_one:      # the reference's treeinfo issues count = 1 per branch
            for k in range(len(mi)):
                if not self.L.pll_update_prob_matrices(
                        self.p, self.params_p,
                        C.cast(mi.ctypes.data + 4 * k, c_uint_p),
                        C.cast(bl.ctypes.data + 8 * k, c_double_p), 1):
                    raise RuntimeError(self.lib.errmsg)
            return
        if not self.L.pll_update_prob_matrices(self.p, self.params_p, mi.ctypes.data_as(c_uint_p),
                                               bl.ctypes.data_as(c_double_p), len(mi)):
            raise RuntimeError(self.lib.errmsg)

    @staticmethod
    def make_ops(op_tuples):
        """op_tuples: (parent_clv, parent_scaler, c1_clv, c1_mat, c1_scaler, c2_clv, c2_mat, c2_scaler)"""
        arr = (Operation * max(1, len(op_tuples)))()
        for k, t in enumerate(op_tuples):
            (arr[k].parent_clv_index, arr[k].parent_scaler_index, arr[k].child1_clv_index,
             arr[k].child1_matrix_index, arr[k].child1_scaler_index, arr[k].child2_clv_index,
             arr[k].child2_matrix_index, arr[k].child2_scaler_index) = [int(x) for x in t]
        return arr

    def update_partials(self, ops, count=None):
        if not isinstance(ops, C.Array):
            count = len(ops)
            ops = self.make_ops(ops)
        self.lib.errno = 0
        self.L.pll_update_partials(self.p, ops, len(ops) if count is None else count)
        if self.lib.errno:
            raise RuntimeError(self.lib.errmsg)

    def edge_lnl(self, pc, psc, cc, csc, matrix, persite=False):
        buf = np.zeros(self.N) if persite else None
        v = self.L.pll_compute_edge_loglikelihood(
            self.p, pc, psc, cc, csc, matrix, self.params_p,
            buf.ctypes.data_as(c_double_p) if persite else None)
        return (v, buf) if persite else v

    def root_lnl(self, clv, sc, persite=False):
        buf = np.zeros(self.N) if persite else None
        v = self.L.pll_compute_root_loglikelihood(
            self.p, clv, sc, self.params_p, buf.ctypes.data_as(c_double_p) if persite else None)
        return (v, buf) if persite else v

    def node_ancestral(self, node, node_sc, other, other_sc, matrix):
        out = np.zeros(self.N * self.S)
        if not self.L.pll_compute_node_ancestral(self.p, node, node_sc, other, other_sc, matrix,
                                                 self.params_p, out.ctypes.data_as(c_double_p)):
            raise RuntimeError(self.lib.errmsg)
        return out.reshape(self.N, self.S)

    def alloc_sumtable(self):
        """caller-owned buffer exactly as src/tree/treeinfo.c:336-340 allocates it"""
        n = self.Nalloc * self.R * self.Sp          # src/tree/treeinfo.c:333-337 adds the AB patterns too
        ptr = self.L.pll_aligned_alloc(max(1, n) * 8, self.p.contents.alignment)
        return C.cast(ptr, c_double_p)

    def free_sumtable(self, st):
        self.L.pll_aligned_free(C.cast(st, C.c_void_p))

    def update_sumtable(self, pc, cc, psc, csc, st):
        if not self.L.pll_update_sumtable(self.p, pc, cc, psc, csc, self.params_p, st):
            raise RuntimeError(self.lib.errmsg)

    def derivatives(self, psc, csc, t, st):
        df, ddf = C.c_double(), C.c_double()
        if not self.L.pll_compute_likelihood_derivatives(self.p, psc, csc, t, self.params_p, st,
                                                         C.byref(df), C.byref(ddf)):
            raise RuntimeError(self.lib.errmsg)
        return df.value, ddf.value

    def newton_branch(self, psc, csc, st, start, bl_min, bl_max, tolerance, max_newton):
        """pllhip_newton_branch: (length, iterations, trail) or raises with the library's message"""
        length, its = C.c_double(), C.c_uint()
        trail = np.zeros(96)
        self.lib.errno = 0
        ok = self.L.pllhip_newton_branch(self.p, psc, csc, self.params_p, st, start, bl_min, bl_max, tolerance,
                                         max_newton, C.byref(length), C.byref(its), trail.ctypes.data_as(c_double_p))
        if not ok:
            raise RuntimeError(f"[{self.lib.errno}] {self.lib.errmsg}")
        return length.value, its.value, trail[:its.value]

    def derivatives_multi(self, psc, csc, ts, st):
        """(df[], ddf[]) at several branch lengths from one sumtable scan"""
        t = _f64(ts)
        df, ddf = np.zeros(len(t)), np.zeros(len(t))
        if not self.L.pllhip_compute_likelihood_derivatives_multi(
                self.p, psc, csc, t.ctypes.data_as(c_double_p), len(t), self.params_p, st,
                df.ctypes.data_as(c_double_p), ddf.ctypes.data_as(c_double_p)):
            raise RuntimeError(self.lib.errmsg)
        return df, ddf

    # --- read-back (works for both libraries) -------------------------------
    def get_clv(self, idx):
        n = self.Nalloc * self.R * self.Sp
        out = np.zeros(n)
        if self.lib.is_product:
            if not self.L.pllhip_get_clv(self.p, idx, out.ctypes.data_as(c_double_p)):
                raise RuntimeError(self.lib.errmsg)
        else:
            ptr = self.p.contents.clv[idx]
            out[:] = np.ctypeslib.as_array(ptr, shape=(n,))
        return out.reshape(self.Nalloc, self.R, self.Sp)[:self.N, :, :self.S]

    def get_scaler(self, idx):
        """scaler counts: [site], or [site][rate] with PLL_ATTRIB_RATE_SCALERS"""
        n = self.Nalloc * (self.R if self.rate_scalers else 1)
        out = np.zeros(n, dtype=np.uint32)
        if self.lib.is_product:
            if not self.L.pllhip_get_scaler(self.p, idx, out.ctypes.data_as(c_uint_p)):
                raise RuntimeError(self.lib.errmsg)
        else:
            out[:] = np.ctypeslib.as_array(self.p.contents.scale_buffer[idx], shape=(n,))
        return out[:self.N * (self.R if self.rate_scalers else 1)]

    def get_pmatrix(self, idx):
        if self.lib.is_product:
            self.L.pllhip_sync_to_host(self.p, PLLHIP_SYNC_PMATRIX)
        n = self.R * self.S * self.Sp
        a = np.ctypeslib.as_array(self.p.contents.pmatrix[idx], shape=(n,)).copy()
        return a.reshape(self.R, self.S, self.Sp)[:, :, :self.S]

    def get_sumtable(self, st):
        n = self.Nalloc * self.R * self.Sp
        out = np.zeros(n)
        if self.lib.is_product:
            if not self.L.pllhip_get_sumtable(self.p, st, out.ctypes.data_as(c_double_p)):
                raise RuntimeError(self.lib.errmsg)
        else:
            out[:] = np.ctypeslib.as_array(st, shape=(n,))
        return out.reshape(self.Nalloc, self.R, self.Sp)[:self.N, :, :self.S]

    def counters(self):
        c = Counters()
        self.L.pllhip_get_counters(self.p, C.byref(c))
        return c

    def repeat_stats(self):
        st = RepeatStats()
        self.L.pllhip_repeat_stats(self.p, C.byref(st))
        return st

    # evaluate-only traversals (include/pllhip.h, pllhip_set_transient)
    def set_transient(self, on=True):
        self.L.pllhip_set_transient(self.p, 1 if on else 0)

    def discard_transient(self):
        self.L.pllhip_discard_transient(self.p)

    def transient_stats(self):
        st = TransientStats()
        self.L.pllhip_transient_stats(self.p, C.byref(st))
        return st


# ---------------------------------------------------------------------------
# synthetic workloads
# ---------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, start=0):
    """values start .. start+n-1 of the splitmix64 stream started at `seed` (vectorised;
    the stream is counter-based, so any range can be generated on its own)"""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _balanced_pick(t):
    """the tip whose pendant edge tip t splits in a balanced tree: tips 0, 1, 2, 3, ... in turn, starting over
    whenever every existing tip has been split once (3 -> 6 -> 12 -> ... tips: a complete binary tree)"""
    start, size = 3, 3
    while t >= start + size:
        start += size
        size *= 2
    return t - start


class Tree:
    """Unrooted binary tree as index arrays.

    tip i -> CLV i; inner node k -> CLV tips+k, scaler k; every edge has a
    unique P-matrix index 0..2n-4.  `ops` is the post-order
    operation list towards the root edge (root_a, root_b, root_matrix).
    """

    def __init__(self, ntips, seed_topology=42, seed_brlen=43, brlen_range=(0.01, 0.2), ladder=False, balanced=False):
        assert ntips >= 3
        self.ntips = ntips
        self.is_ladder = ladder
        rnd = splitmix64(seed_topology, ntips)
        # edges as [u, v]; start with a star on tips 0,1,2 around inner node n
        edges = [[0, ntips], [1, ntips], [2, ntips]]
        pendant = {0: 0, 1: 1, 2: 2}            # tip -> index of its pendant edge
        for t in range(3, ntips):
            # ladder: always split the pendant edge of the tip added last (a caterpillar);
            # balanced: the pendant edges of the tips in turn (every tip becomes a cherry before any is split twice)
            if ladder:
                e = len(edges) - 1
            elif balanced:
                e = pendant[_balanced_pick(t)]
            else:
                e = int(rnd[t] % np.uint64(len(edges)))
            u, v = edges[e]
            w = ntips + (t - 2)          # new inner node
            edges[e] = [u, w]
            edges.append([w, v])
            edges.append([t, w])
            if u < ntips:                # the split edge was the pendant edge of tip u: it still is (u -- w)
                pendant[u] = e
            elif v < ntips:
                pendant[v] = len(edges) - 2
            pendant[t] = len(edges) - 1
        self.edges = edges
        lo, hi = brlen_range
        self.brlens = lo + (hi - lo) * uniform01(seed_brlen, len(edges))
        self.nedges = len(edges)
        adj = {}
        for k, (u, v) in enumerate(edges):
            adj.setdefault(u, []).append((v, k))
            adj.setdefault(v, []).append((u, k))
        self.adj = adj
        self.set_root_edge(self.nedges - 1 if ntips > 3 else 2)

    def scaler_of(self, node):
        return node - self.ntips if node >= self.ntips else PLL_SCALE_BUFFER_NONE

    def _postorder(self, node, parent, out):
        """iterative post-order of the subtree at `node` seen from `parent`"""
        stack = [(node, parent, False)]
        while stack:
            nd, par, done = stack.pop()
            if nd < self.ntips:
                continue
            kids = [(v, k) for (v, k) in self.adj[nd] if v != par]
            if done:
                (c1, m1), (c2, m2) = kids
                out.append((nd, self.scaler_of(nd), c1, m1, self.scaler_of(c1),
                            c2, m2, self.scaler_of(c2)))
            else:
                stack.append((nd, par, True))
                for v, _ in kids:
                    stack.append((v, nd, False))

    def set_root_edge(self, k):
        u, v = self.edges[k]
        if u < self.ntips:       # keep an inner node on the "parent" side
            u, v = v, u
        self.root_a, self.root_b, self.root_matrix = u, v, k
        ops = []
        self._postorder(v, u, ops)
        self._postorder(u, v, ops)
        self.ops = ops

    def ops_with_scalers(self, use):
        if use:
            return self.ops
        N = PLL_SCALE_BUFFER_NONE
        return [(p, N, a, ma, N, b, mb, N) for (p, _, a, ma, _, b, mb, _) in self.ops]

    def newick(self, labels=None):
        def rec(nd, par, k):
            if nd < self.ntips:
                s = labels[nd] if labels else f"t{nd}"
            else:
                s = "(" + ",".join(rec(v, nd, kk) for (v, kk) in self.adj[nd] if v != par) + ")"
            return s + (f":{self.brlens[k]:.17g}" if k is not None else "")
        root = self.ntips
        return "(" + ",".join(rec(v, root, kk) for (v, kk) in self.adj[root]) + ");"


DNA_GTR_RATES = [1.452176, 0.937951, 0.462880, 0.617729, 1.745312, 1.0]   # blopt-minimal.c:49
DNA_FREQS = [0.17, 0.19, 0.25, 0.39]                                       # spr-round.c:153


def protein_model(seed_rates=46, seed_freqs=47):
    """'LG-shaped' reversible 20-state model: 190 log-normal exchangeabilities
    and Dirichlet(5) frequencies.  The real LG table is not available offline
    (src/util/models_aa.c:29 takes it from libpll), so this is NOT LG."""
    u = uniform01(seed_rates, 380)
    z = np.sqrt(-2.0 * np.log(np.maximum(u[0::2], 1e-300))) * np.cos(2 * np.pi * u[1::2])
    rates = np.exp(z)
    rates[-1] = 1.0
    g = -np.log(np.maximum(uniform01(seed_freqs, 100), 1e-300)).reshape(20, 5).sum(axis=1)
    return rates, g / g.sum()


def codon_model(kappa=2.0, omega=0.2, seed_freqs=48):
    """GY94-shaped 61-state model from the standard genetic code."""
    bases = "TCAG"
    aa = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"
    codons = [(a + b + c) for a in bases for b in bases for c in bases]
    sense = [(cd, aa[i]) for i, cd in enumerate(codons) if aa[i] != "*"]
    n = len(sense)
    assert n == 61
    transitions = {("T", "C"), ("C", "T"), ("A", "G"), ("G", "A")}
    rates = []
    for i in range(n):
        for j in range(i + 1, n):
            ci, ai = sense[i]
            cj, aj = sense[j]
            diff = [(x, y) for x, y in zip(ci, cj) if x != y]
            if len(diff) != 1:
                rates.append(0.0)
                continue
            r = 1.0
            if diff[0] in transitions:
                r *= kappa
            if ai != aj:
                r *= omega
            rates.append(r)
    g = -np.log(np.maximum(uniform01(seed_freqs, 61 * 5), 1e-300)).reshape(61, 5).sum(axis=1)
    return np.array(rates), g / g.sum()


def random_codes(ntips, nsites, nstates, seed=44, first_site=0):
    """iid uniform unambiguous state index per (tip, site) -> uint8 [tips][sites];
    sites first_site .. first_site+nsites-1 of the alignment that `seed` defines (a
    rank of a multi-GPU run generates just its slice)"""
    out = np.empty((ntips, nsites), dtype=np.uint8)
    for t in range(ntips):
        out[t] = (splitmix64(seed + 1000003 * t, nsites, first_site) % np.uint64(nstates)).astype(np.uint8)
    return out


def simulated_codes(tree, nsites, nstates, seed=45, scale=1.0):
    """states evolved along `tree` (a Tree) from inner node `ntips` outwards:
    equal-rates process, two site-rate classes; uint8 [tips][sites].  Gives an
    alignment with phylogenetic signal (seed 45)."""
    out = np.empty((tree.ntips, nsites), dtype=np.uint8)
    S = nstates
    rate = np.where(uniform01(seed, nsites) < 0.5, 0.4, 1.6)
    start = (splitmix64(seed + 1, nsites) % np.uint64(S)).astype(np.int64)
    stack = [(tree.ntips, -1, start)]
    while stack:
        node, parent, state = stack.pop()
        if node < tree.ntips:
            out[node] = state.astype(np.uint8)
            continue
        for child, k in tree.adj[node]:
            if child == parent:
                continue
            t = tree.brlens[k] * scale
            pchange = (S - 1.0) / S * (1.0 - np.exp(-S / (S - 1.0) * t * rate))
            u = uniform01(seed + 7919 * (k + 1), nsites)
            jump = 1 + (splitmix64(seed + 104729 * (k + 1), nsites) % np.uint64(S - 1)).astype(np.int64)
            stack.append((child, node, np.where(u < pchange, (state + jump) % S, state)))
    return out


def state_charmap(nstates):
    """a 256-entry char -> mask map where byte value 48+i ('0'+i) means state i and
    '-' means every state (so uint8 state arrays + 48 are valid sequences)"""
    m = np.zeros(256, dtype=np.uint64)
    for i in range(nstates):
        m[48 + i] = np.uint64(1) << np.uint64(i)
    m[ord("-")] = (np.uint64(1) << np.uint64(nstates)) - np.uint64(1) if nstates < 64 else _M64
    return m


class SprParams(C.Structure):
    """pllhip_spr_params_t, include/pllhip_eval.h"""
    _fields_ = [("radius_min", C.c_uint), ("radius_max", C.c_uint), ("ntopol_keep", C.c_uint),
                ("thorough", C.c_int), ("bl_min", C.c_double), ("bl_max", C.c_double),
                ("smoothings", C.c_int), ("epsilon", C.c_double), ("subtree_cutoff", C.c_double),
                ("lh_epsilon_brlen_triplet", C.c_double)]


class SprCutoff(C.Structure):
    _fields_ = [("lh_start", C.c_double), ("lh_cutoff", C.c_double), ("lh_dec_sum", C.c_double),
                ("lh_dec_count", C.c_int)]


SPR_LOG_MAX = 256


class SprStats(C.Structure):
    _fields_ = [("prunings", C.c_ulong), ("insertions", C.c_ulong), ("moves_applied", C.c_ulong),
                ("rescored", C.c_ulong), ("lnl_start", C.c_double), ("lnl_scan", C.c_double),
                ("lnl_final", C.c_double), ("log_count", C.c_uint),
                ("log_prune", C.c_uint * SPR_LOG_MAX), ("log_regraft", C.c_uint * SPR_LOG_MAX)]


class Evaluation:
    """pll_utree_t + partitions + pllhip_eval driver (include/pllhip_eval.h).

    The tree comes from a newick string; tips are matched to alignment rows by
    their labels "t<k>" (the parser numbers tips in order of appearance)."""

    def __init__(self, lib, newick, flags=0, nparts=1):
        self.lib, self.L = lib, lib.lib
        self.utree = self.L.pll_utree_parse_newick_string(newick.encode())
        if not self.utree:
            raise RuntimeError(lib.errmsg)
        tr = self.utree.contents
        self.ntips = tr.tip_count
        # row k of an alignment belongs to the tip labelled t<k>
        self.tip_clv = {}
        for i in range(self.ntips):
            nd = tr.nodes[i].contents
            self.tip_clv[int(nd.label.decode()[1:])] = nd.clv_index
        self.ev = self.L.pllhip_eval_create(self.utree, nparts, flags)
        if not self.ev:
            raise RuntimeError(lib.errmsg)
        self.parts = []

    def add_partition(self, index, states, nsites, rate_cats, codes, subst, freqs, alpha, coded=True,
                      attributes=0):
        inst = Instance(self.lib, self.ntips, states, nsites, rate_cats,
                        attributes=(PLL_ATTRIB_PATTERN_TIP if coded else 0) | attributes)
        rates = self.lib.gamma_cats(alpha, rate_cats) if rate_cats > 1 else np.ones(1)
        inst.set_model(subst, freqs, rates)
        cmap = state_charmap(states)
        for k in range(self.ntips):
            inst.set_tip_states(self.tip_clv[k], cmap, (codes[k] + 48).tobytes())
        if not self.L.pllhip_eval_set_partition(self.ev, index, inst.p, inst.params_p):
            raise RuntimeError(self.lib.errmsg)
        self.parts.append(inst)
        return inst

    def add_remote_partition(self, index):
        """a partition another worker owns: a NULL slot (src/tree/treeinfo.c:1024-1031)"""
        if not self.L.pllhip_eval_set_partition(self.ev, index, None, None):
            raise RuntimeError(self.lib.errmsg)

    def set_parallel_context(self, cb, ctx=None):
        """cb: a REDUCE_CB instance (kept alive here) or a raw function pointer"""
        self._cb = cb
        self.L.pllhip_eval_set_parallel_context(self.ev, ctx, C.cast(cb, C.c_void_p))

    def attach_comm(self, comm=None):
        """deferred results (device-side reduce); product library only"""
        if not self.L.pllhip_eval_attach_comm(self.ev, comm):
            raise RuntimeError(self.lib.errmsg)

    def records(self):
        """every node record of the tree (pll_unode_t pointers)"""
        tr = self.utree.contents
        for i in range(tr.tip_count + tr.inner_count):
            n = tr.nodes[i]
            s_ = n
            while True:
                yield s_
                if not s_.contents.next:
                    break
                s_ = s_.contents.next
                if C.addressof(s_.contents) == C.addressof(n.contents):
                    break

    def set_linkage(self, linkage, scalers=None):
        """0 linked, 1 scaled (per-partition scalers), 2 unlinked (per-partition lengths =
        tree length x scalers[p] to start with)"""
        if not self.L.pllhip_eval_set_brlen_linkage(self.ev, linkage):
            raise RuntimeError(self.lib.errmsg)
        for p, sc in enumerate(scalers or []):
            if linkage == 1:
                if not self.L.pllhip_eval_set_brlen_scaler(self.ev, p, sc):
                    raise RuntimeError(self.lib.errmsg)
            elif linkage == 2:
                for rec in self.records():
                    if not self.L.pllhip_eval_set_partition_branch_length(self.ev, p, rec, rec.contents.length * sc):
                        raise RuntimeError(self.lib.errmsg)

    def partition_tree_length(self, p):
        tot = 0.0
        for rec in self.records():
            tot += self.L.pllhip_eval_get_partition_branch_length(self.ev, p, rec) * \
                self.L.pllhip_eval_get_brlen_scaler(self.ev, p) / 2
        return tot

    def newton_iterations(self):
        return self.L.pllhip_eval_newton_iterations(self.ev)

    def root_edge(self):
        """(clv, scaler, back clv, back scaler, pmatrix) of the edge the lnL is computed at"""
        r = self.L.pllhip_eval_root(self.ev).contents
        b = r.back.contents
        return r.clv_index, r.scaler_index, b.clv_index, b.scaler_index, r.pmatrix_index

    def persite_lnl(self, index=0):
        """per-site lnL of partition `index` at the root edge (CLVs must be valid: call loglh first)"""
        pc_, psc, cc, csc, m = self.root_edge()
        return self.parts[index].edge_lnl(pc_, psc, cc, csc, m, persite=True)

    def set_transient(self, mode):
        """0 off, 1 every full evaluation, 2 a full evaluation that directly follows one (include/pllhip_eval.h)"""
        self.L.pllhip_eval_set_transient(self.ev, int(mode))

    def loglh(self, incremental=False):
        v = self.L.pllhip_eval_loglh(self.ev, 1 if incremental else 0)
        if v != v:
            raise RuntimeError(self.lib.errmsg)
        return v

    def optimize_branches(self, bl_min=1e-4, bl_max=10.0, eps=0.01, iters=8, radius=-1):
        self.lib.errno = 0
        v = self.L.pllhip_eval_optimize_branches(self.ev, bl_min, bl_max, eps, iters, radius)
        if v == 0.0 or self.lib.errno:
            raise RuntimeError(f"[{self.lib.errno}] {self.lib.errmsg}")
        return -v

    def spr_round(self, radius_min=1, radius_max=5, ntopol_keep=5, thorough=False, bl_min=1e-4,
                  bl_max=10.0, smoothings=8, epsilon=0.1, subtree_cutoff=1.0, triplet_epsilon=0.1,
                  cutoff=None):
        """one SPR round; returns (lnL, SprStats); `cutoff` (SprCutoff) is carried between rounds"""
        prm = SprParams(radius_min, radius_max, ntopol_keep, int(thorough), bl_min, bl_max, smoothings,
                        epsilon, subtree_cutoff, triplet_epsilon)
        st = SprStats()
        self.lib.errno = 0
        v = self.L.pllhip_eval_spr_round(self.ev, C.byref(prm), C.byref(cutoff) if cutoff is not None else None,
                                         C.byref(st))
        if v == 0.0:
            raise RuntimeError(f"[{self.lib.errno}] {self.lib.errmsg}")
        return v, st

    def newick(self):
        root = self.L.pllhip_eval_root(self.ev)
        ptr = self.L.pll_utree_export_newick(root, None)
        text = C.string_at(ptr).decode()
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(ptr)
        return text

    def counters(self):
        return (self.L.pllhip_eval_ops(self.ev), self.L.pllhip_eval_pmatrix_updates(self.ev),
                self.L.pllhip_eval_derivative_calls(self.ev))

    def close(self):
        if self.ev:
            self.L.pllhip_eval_destroy(self.ev)
            self.ev = None
        for p in self.parts:
            p.close()
        self.parts = []
        if self.utree:
            self.L.pll_utree_destroy(self.utree, None)
            self.utree = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


CONFIGS = {
    # name: (states, rate_cats, taxa, sites)   BASELINE.json configs
    "c1": (4, 1, 10, 1000),
    "c2": (4, 4, 100, 1_000_000),
    "c3": (20, 4, 200, 1_000_000),
    "c4": (20, 4, 100, 1_000_000),     # site count split 25/25/12.5/12.5 % over 2 DNA + 2 protein partitions
    "c5": (61, 4, 50, 200_000),
}


def mixture_component(subst, freqs, m):
    """model of mixture component m > 0: the base model's exchangeabilities and frequencies under
    seeded log-normal / Gamma noise (component 0 is the base model itself)"""
    subst, freqs = np.asarray(subst, dtype=float), np.asarray(freqs, dtype=float)
    if m == 0:
        return subst, freqs
    u = uniform01(9000 + 17 * m, 2 * len(subst))
    z = np.sqrt(-2.0 * np.log(np.maximum(u[0::2], 1e-300))) * np.cos(2 * np.pi * u[1::2])
    sub = subst * np.exp(0.4 * z)
    g = freqs * (0.5 + uniform01(9500 + 17 * m, len(freqs)))
    return sub, g / g.sum()


def build_instance(lib, states, rate_cats, ntips, nsites, coded=True, scalers=True, alpha=None,
                   seed_shift=0, tree=None, pinv=0.0, attributes=0, mixture=None, mixture_pinv=None, codes=None):
    """partition + tree + model + tips for one synthetic configuration.
    mixture: per-rate-category rate-matrix indices (e.g. [0, 1, 0, 1]): the partition gets
    max(mixture) + 1 rate matrices, each with a model of its own (mixture_component) and, with
    mixture_pinv, its own proportion of invariant sites."""
    tree = tree or Tree(ntips, 42 + seed_shift, 43 + seed_shift)
    nrm = 1 if mixture is None else int(max(mixture)) + 1
    inst = Instance(lib, ntips, states, nsites, rate_cats,
                    attributes=(PLL_ATTRIB_PATTERN_TIP if coded else 0) | attributes, scalers=scalers,
                    rate_matrices=nrm)
    if states == 4:
        subst, freqs, a = DNA_GTR_RATES, DNA_FREQS, 0.841
    elif states == 20:
        (subst, freqs), a = protein_model(), 0.5
    elif states == 61:
        (subst, freqs), a = codon_model(), 0.5
    else:
        ns = states * (states - 1) // 2
        subst = 0.5 + uniform01(146 + states, ns)
        f = 0.5 + uniform01(147 + states, states)
        freqs, a = f / f.sum(), 0.7
    alpha = a if alpha is None else alpha
    rates = lib.gamma_cats(alpha, rate_cats) if rate_cats > 1 else np.ones(1)
    inst.set_model(subst, freqs, rates)
    if mixture is not None:
        assert len(mixture) == rate_cats
        for m in range(1, nrm):
            inst.set_model(*mixture_component(subst, freqs, m), rates, idx=m)
        inst.set_params_indices(mixture)
    cmap = state_charmap(states)
    if codes is None:
        codes = random_codes(ntips, nsites, states, 44 + seed_shift)
    for t in range(ntips):
        inst.set_tip_states(t, cmap, (codes[t] + 48).tobytes())
    if pinv > 0:
        inst.set_pinv(pinv)
    if mixture_pinv is not None:
        if not inst.L.pll_update_invariant_sites(inst.p):
            raise RuntimeError(lib.errmsg)
        for m, v in enumerate(mixture_pinv):
            inst.set_pinv(v, idx=m)
    inst.tree = tree
    inst.codes = codes
    return inst


def newton_branch_multi(lib, insts, psc, csc, sumtables, scalers, start, bl_min, bl_max, tolerance, max_newton):
    """pllhip_newton_branch_multi over `insts` (their own sumtables, optional branch-length scalers):
    (length, iterations, trail) or raises with the library's message"""
    n = len(insts)
    parts = (C.POINTER(Partition) * n)(*[i.p for i in insts])
    params = (c_uint_p * n)(*[i.params_p for i in insts])
    sts = (c_double_p * n)(*[C.cast(s_, c_double_p) for s_ in sumtables])
    sc = _f64(scalers) if scalers is not None else None
    length, its = C.c_double(), C.c_uint()
    trail = np.zeros(96)
    lib.errno = 0
    ok = lib.lib.pllhip_newton_branch_multi(parts, n, psc, csc, params, sts, sc.ctypes.data_as(c_double_p) if sc is not None else None,
                                            start, bl_min, bl_max, tolerance, max_newton, C.byref(length), C.byref(its),
                                            trail.ctypes.data_as(c_double_p))
    if not ok:
        raise RuntimeError(f"[{lib.errno}] {lib.errmsg}")
    return length.value, its.value, trail[:its.value]


def update_partials_batch(lib, insts, ops, count=None):
    """pllhip_update_partials_batch over `insts` (None = a partition another worker owns)"""
    if not isinstance(ops, C.Array):
        count = len(ops)
        ops = Instance.make_ops(ops)
    arr = (C.POINTER(Partition) * len(insts))(*[i.p if i is not None else None for i in insts])
    lib.errno = 0
    if not lib.lib.pllhip_update_partials_batch(arr, len(insts), ops, len(ops) if count is None else count):
        raise RuntimeError(lib.errmsg)


def full_traversal(inst, one_by_one_pmatrices=False):
    """the W1 workload: all P-matrices, n-2 partial ops, one edge lnL
    (call pattern of treeinfo_compute_loglh, src/tree/treeinfo.c:946-1079)"""
    t = inst.tree
    key = (id(t), t.root_matrix)
    if getattr(inst, "_trav_key", None) != key:      # C arrays built once per (tree, root)
        inst._trav_key = key
        inst._trav_mi = _u32(np.arange(t.nedges))
        inst._trav_ops = inst.make_ops(t.ops_with_scalers(inst.nscalers > 0))
        inst._trav_nops = len(t.ops)
    if one_by_one_pmatrices:
        # one call per branch, as treeinfo issues them (src/tree/treeinfo.c:845-865);
        # the argument pointers are prepared once so that the loop is just the calls
        if getattr(inst, "_trav_ptrs", None) is None or inst._trav_ptrs[0] != key:
            bl = inst._trav_bl = _f64(t.brlens)
            inst._trav_ptrs = (key, [(C.cast(inst._trav_mi.ctypes.data + 4 * k, c_uint_p),
                                      C.cast(bl.ctypes.data + 8 * k, c_double_p))
                                     for k in range(t.nedges)])
        inst._trav_bl[:] = t.brlens
        fn, p, pp = inst.L.pll_update_prob_matrices, inst.p, inst.params_p
        for mi_p, bl_p in inst._trav_ptrs[1]:
            if not fn(p, pp, mi_p, bl_p, 1):
                raise RuntimeError(inst.lib.errmsg)
    else:
        bl = _f64(t.brlens)
        if not inst.L.pll_update_prob_matrices(inst.p, inst.params_p,
                                               inst._trav_mi.ctypes.data_as(c_uint_p),
                                               bl.ctypes.data_as(c_double_p), t.nedges):
            raise RuntimeError(inst.lib.errmsg)
    inst.update_partials(inst._trav_ops, inst._trav_nops)
    sa = t.scaler_of(t.root_a) if inst.nscalers else PLL_SCALE_BUFFER_NONE
    sb = t.scaler_of(t.root_b) if inst.nscalers else PLL_SCALE_BUFFER_NONE
    return inst.edge_lnl(t.root_a, sa, t.root_b, sb, t.root_matrix)
