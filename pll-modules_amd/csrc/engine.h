// engine.h -- internal state of the HIP likelihood engine behind include/pll.h.
//
// Ownership model (SURVEY.md section 8b "Ownership"):
//   * CLVs, scalers, tip codes, pattern weights, P-matrices, tip lookup tables
//     and sumtables live in HBM for the lifetime of the partition.
//   * The small model arrays of pll_partition_t (rates, rate_weights,
//     frequencies, prop_invar, eigen*) stay on the host as the source of truth;
//     every kernel-entry call compares them with a shadow copy and re-uploads
//     on change, because pll-modules writes them without calling a setter
//     (src/algorithm/algo_callback.c:44-68, 93-118).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <list>
#include <utility>

#include "pll.h"
#include "pllhip.h"

namespace pllhip {

constexpr unsigned MAX_RATE_CATS = 16;   // params_indices travel by value in kernel args
constexpr unsigned MAX_OPS_PER_LAUNCH = 24;
constexpr unsigned MAX_PMAT_PER_LAUNCH = 200;   // requests of one k_pmatrix launch (by value in the kernel arguments)
constexpr unsigned REDUCE_BLOCKS = 4096; // upper bound of per-block partial sums
constexpr unsigned REDUCE_QUANTITIES = 16; // sums per launch: up to 8 trial branch lengths x {df, ddf}
constexpr unsigned MAX_TRIAL_LENGTHS = 8;
constexpr unsigned REDUCE_COUNTER_WORDS = 8 * 1024 + 1;  // kernels_common.hpp: 8 shard tickets 4 KiB apart + top ticket
constexpr unsigned RESULT_WORDS = 128;    // mapped result buffer: 16 values, sequence word at RESULT_SEQ_SLOT, AB tail
constexpr unsigned RESULT_ASC_SLOT = 32;  // log-likelihoods of the <= 64 ascertainment-bias columns
constexpr unsigned RESULT_SEQ_SLOT = 24;
constexpr unsigned MAX_SUMTABLES = 4;    // device sumtables kept per partition (LRU)

void set_error(int code, const char * fmt, ...);

// hip error -> pll_errno; returns true on success
bool hip_ok(hipError_t e, const char * what);
#define PLLHIP_TRY(call) do { if (!::pllhip::hip_ok((call), #call)) return PLL_FAILURE; } while (0)

// One pruning step with every index already resolved to a device pointer.
// A child is either an inner/full CLV (clv != nullptr) or a coded tip
// (codes != nullptr, lut = per-matrix lookup table).
struct OpDesc
{
  const double * clv1;  const uint8_t * codes1;  const double * pmat1;  const double * lut1;
  const double * clv2;  const uint8_t * codes2;  const double * pmat2;  const double * lut2;
  const unsigned * scaler1;
  const unsigned * scaler2;
  double * parent;
  unsigned * parent_scaler;
  const double * pfrag1, * pfrag2;   // 20 states: the children's matrices in fragment order (or null)
  unsigned parent_index;      // CLV indices (host bookkeeping of per-vector side arrays)
  unsigned child1_index, child2_index;
};

// 61-state rate-parallel launches: predicted per-site scaling decision of every parent
// vector (what the last evaluation of that vector decided), and where the new one goes
struct PredBatch
{
  const uint8_t * in[MAX_OPS_PER_LAUNCH];
  uint8_t * out[MAX_OPS_PER_LAUNCH];
};

// trial branch lengths of one derivative launch, by value in kernel arguments
struct TrialLengths { double t[MAX_TRIAL_LENGTHS]; };

// per-rate parameter-set indices, by value in kernel arguments
struct ParamIdxHost { unsigned v[16]; };

struct OpBatch
{
  OpDesc op[MAX_OPS_PER_LAUNCH];
};

// Operation chains (20-state family): consecutive operations along one root-ward
// path of the tree, run by ONE launch; the parent vector of a link is handed to the
// next operation in registers, so that operation never reads it back from HBM.
struct ChainBatch
{
  OpDesc op[MAX_OPS_PER_LAUNCH];                 // chains back to back
  unsigned char first[MAX_OPS_PER_LAUNCH];       // per chain: index of its first op
  unsigned char len[MAX_OPS_PER_LAUNCH];         // per chain: number of ops
  unsigned char carried[MAX_OPS_PER_LAUNCH];     // per op: 0 = both children from memory,
                                                 // 1 / 2 = child 1 / 2 is the previous op's parent
  unsigned short slot1[MAX_OPS_PER_LAUNCH];      // per op (20 states): LDS offsets, in doubles from the
  unsigned short slot2[MAX_OPS_PER_LAUNCH];      // start of its chain's area, of the two children's tables
};

// Whole traversals in ONE launch.  Felsenstein pruning never mixes sites, so a wave that keeps
// the same site blocks through every operation of a traversal depends on nobody else: the
// schedule (chains in dependency order) lives in device memory, a workgroup walks ALL its chains
// -- re-staging its LDS tables between two chains -- and no kernel boundary (launch gap, L2
// write-back, refill of every CU) separates the rounds of chains any more.
struct PlanOp
{
  OpDesc d;
  unsigned carried;      // 0 = both children from memory, 1 / 2 = child 1 / 2 is the previous op's parent
  unsigned slot1, slot2; // 20 states: LDS offsets (doubles) of the two children's tables
  unsigned flags;        // bit 0: the parent vector is not stored (an evaluate-only traversal hands it to the next
                         // operation of its chain in registers; pllhip_set_transient)
};
// operations [first, first + len) of PlanOp[], and what the kernels need to know about the partition
// the chain belongs to (a batched schedule -- pllhip_update_partials_batch -- holds chains of several
// partitions): extent = site blocks (blocked families) or sites (4-state family), the row count of its
// tip lookup tables in memory, the rows staged in LDS, flags (bit 0: tip tables are staged in LDS)
struct PlanChain { unsigned first, len, extent, lut_codes, lut_used, flags; };
// A schedule of a few operations (the single-operation updates of a branch-length pass or of an SPR insertion) travels
// in the kernel arguments instead of being copied to the device first: inline_ops != 0 = bytes of the PlanOp array at the
// start of `inl`, the PlanChain array behind it (the kernels read both through the kernel-argument segment: plan_bases)
constexpr unsigned PLAN_INLINE_BYTES = 1024;
struct PlanView
{
  const PlanOp * ops;
  const PlanChain * chains;
  unsigned nchains, inline_ops;
  unsigned long long inl[PLAN_INLINE_BYTES / 8];
};

struct DevicePlan                                 // the schedule resident on the device
{
  unsigned char * d_buf = nullptr;                // [PlanOp x nops][PlanChain x nchains]
  size_t cap = 0;
  unsigned char * h_stage = nullptr;              // pinned staging of the upload
  size_t h_cap = 0;
  hipEvent_t copied = nullptr;                    // the last upload has left h_stage
  std::vector<unsigned char> resident;            // what d_buf holds (an unchanged schedule is not uploaded again)
  // the schedule of the last traversal, kept so that a repeated operation list is not planned again
  std::vector<unsigned char> key;                 // operation list + settings it was built from
  std::vector<unsigned char> bytes;               // serialised [PlanOp ...][PlanChain ...]
  unsigned nops = 0, nchains = 0, lds_doubles = 0;
  double algo_bytes = 0.0, algo_flops = 0.0;      // algorithmic traffic / work of the traversal
  double min_bytes = 0.0;                         // traffic without the child vectors handed over in registers
  // launches of the schedule: one for a whole traversal, or one per round of chains (chains
  // [begin, end) side by side) for a partition that does not fill the chip on its own
  struct Launch { unsigned begin, end, rows, ops; double bytes, flops, min_bytes; };   // chains [begin, end) over `rows` grid rows
  std::vector<Launch> launches;
  unsigned long long generation = 0;              // changes whenever `bytes` / `launches` are rebuilt (unique across engines)
  // site repeats: the cherries the schedule keeps per class and the lookup tables it reads them through
  // (job arrays behind the chains in `bytes`)
  unsigned ncherry_jobs = 0, npair_jobs = 0, repeat_codes = 0;
  size_t off_cherry_jobs = 0, off_pair_jobs = 0;
  // the class operations run level by level (a node's table needs the tables below it): per level the row tables
  // of its class children (pair jobs), then the tables (class jobs); the last entry holds the row tables of the
  // wide tips of the chains (no class jobs)
  struct RepeatLevel { unsigned job_begin, job_end, pair_begin, pair_end, max_classes, max_rows; };
  std::vector<RepeatLevel> repeat_levels;
  unsigned long long repeat_classes = 0;          // classes of all class operations (statistics)
  unsigned max_extent = 0;                        // largest PlanChain::extent of the schedule
};

// a schedule over several partitions of one kernel family (pllhip_update_partials_batch): the members'
// round schedules side by side, resident in the leader's memory
struct BatchPlan
{
  std::vector<const void *> members;              // engines, in call order
  std::vector<unsigned long long> generations;    // of the members' schedules it was merged from
  DevicePlan plan;
  std::vector<hipEvent_t> ready;                  // per member: its stream has reached the batch call
  hipEvent_t done = nullptr;                      // the leader's stream has run the batch
};

enum class KernelFamily { Generic, S4, S16, S20, S61 };

struct Engine
{
  int device = 0;
  hipStream_t stream = nullptr;
  bool counted = false;               // in the per-device count of live partitions
  // engine-internal sharding (include/pllhip.h, pllhip_set_sharding): a ROUTER owns no device
  // memory; it forwards every call to its shards, ordinary partitions that each hold a
  // contiguous site range on their own device
  std::vector<pll_partition_t *> shards;
  std::vector<unsigned> shard_first;   // first site of every shard (+ the total at the end)

  unsigned S = 0, Sp = 0, R = 0, N = 0;   // N: patterns the arrays hold (alignment + ascertainment-bias columns)
  unsigned Nreal = 0;                 // alignment patterns (= partition->sites); N - Nreal = S with AB, else 0
  double * h_asc = nullptr;           // pinned + mapped, AB only: per trial length and state {A, B, C, count}
  double * d_asc = nullptr;           // device view of h_asc
  unsigned tips = 0, nodes = 0, nscalers = 0, nmat = 0, nrm = 0;
  bool coded_tips = false;
  bool rate_scalers = false;          // PLL_ATTRIB_RATE_SCALERS: one count per (site, rate), scaler[n*R + r]
  // PLL_ATTRIB_SITE_REPEATS, first step (kernels_repeats.hpp): a cherry is kept per class of sites (pair of
  // tip codes) and expanded to the site-indexed vector only for a reader that needs it
  bool site_repeats = false;
  bool tip_classes = false;           // tips that are vectors are kept per class of sites as well (upload_tip_classes)
  // 33 .. 64 states without PLL_ATTRIB_PATTERN_TIP: tips given through pll_set_tip_states keep byte codes next to their
  // vectors (a code table of the engine's own: the state masks met so far), and the operations above them read them
  // as coded tips -- lookup tables instead of a 64 x 64 matrix product per site and rate (upload_tip_classes)
  bool shadow_codes = false;
  std::vector<unsigned long long> shadow_tipmap;
  std::vector<char> tip_has_codes;    // the tip's codes are those of its vector (not after pll_set_tip_clv / pllhip_set_clv)
  struct Cherry                       // a node known per class of sites (a cherry, or a node above class nodes / tips)
  {
    bool valid = false;               // the node's vector IS this table (nothing has overwritten it since)
    bool materialized = false;        // d_clv[node] holds the expanded vector
    unsigned ncodes = 0;              // size of the tip code table the class map was made under
    unsigned nclasses = 0;
    double * table = nullptr;         // blocked pseudo-CLV over the classes
    unsigned * pair = nullptr;        // [Nalloc] class per site
    uint8_t * flags = nullptr;        // [classes] scaled?
    unsigned * counts = nullptr;      // [classes] scaler count of the class: own decision + the counts of the children's classes
    int scaler_index = -1;            // the scale buffer the class operation was given (PLL_SCALE_BUFFER_NONE = -1)
    unsigned cap_classes = 0;         // what table / flags were allocated for
    unsigned * rep = nullptr;         // [2 * classes] the children's classes of each class (null: a cherry, class = code1 * ncodes + code2)
    unsigned rep_cap = 0;
    // the class map (pair, rep, nclasses) depends on the topology below and the tips' codes only: kept while
    // the children and their maps are the ones it was made from
    bool map_valid = false;
    bool trackable = false;           // map_valid: the node is worth keeping per class (few enough classes)
    unsigned child[2] = {~0u, ~0u};
    unsigned long long child_version[2] = {0, 0};
    unsigned long long version = 0;   // changes with the map
  };
  std::vector<Cherry> cherries;       // by CLV index (empty unless site_repeats)
  // scale buffers whose per-site counts exist per class only: scaler_lazy[buffer] = the class node whose `counts`
  // stand for it (-1: the buffer holds its counts per site); need_scaler() writes them out for a reader
  std::vector<int> scaler_lazy;
  std::vector<unsigned long long> tip_version;    // changes with a tip's codes
  unsigned long long class_clock = 0;
  unsigned * d_class_seen = nullptr;  // [pairs of child classes] scratch of the class numbering, + tile sums + total
  size_t class_seen_cap = 0;
  unsigned * h_class_total = nullptr; // pinned: the class count read back once per new map
  double * d_pairlut = nullptr;       // lookup tables of the wide tips of the resident schedule
  size_t pairlut_cap = 0;
  pllhip_repeat_stats_t repeat_stats = {};
  // Evaluate-only traversals (pllhip_set_transient): a resident schedule may hand the vectors inside its chains on
  // in registers without storing them.  Such a vector stays recomputable -- the operation that made it is kept --
  // and is stored for the first reader that needs it, or before one of its inputs changes.
  pll_partition_t * owner = nullptr;
  bool transient_mode = false;
  bool transient_busy = false;                  // a materialisation is running (its lists are exempt from the checks)
  unsigned ntransient = 0;                      // vectors that exist as their operation only
  std::vector<char> transient_live;             // [nodes]
  std::vector<pll_operation_t> transient_op;    // [nodes] what recomputes the vector
  std::vector<unsigned> transient_mat_users;    // [nmat] live operations that read the matrix
  pllhip_transient_stats_t transient_stats = {};
  size_t sc_len = 0;                  // entries per scale buffer on the device
  KernelFamily family = KernelFamily::Generic;
  unsigned cu_count = 256;
  // S20 family: CLVs/sumtables live in 32-site blocks [block][rate][state][32]
  // (kernels_s20.hpp); per-site arrays are padded to whole blocks
  bool blocked = false;
  unsigned nblk = 0;                  // site blocks (blocked layout only)
  unsigned rows = 0;                  // state rows per blocked unit (>= S, a multiple of 4; 0: API layout)
  size_t clv_len = 0;                 // doubles per CLV / sumtable buffer
  unsigned Nalloc = 0;                // per-site array length (N, or nblk*32)
  double * d_sum_scratch = nullptr;   // eigen-basis matrices + LUTs of the sumtable kernel
  // ... which depend on the model, the parameter indices and the code table only, not on the branch: what the
  // scratch area was last prepared for (sum_prep_needed)
  unsigned long long model_generation = 0;          // uploads of the model block
  unsigned long long sum_prep_model = ~0ULL;
  unsigned sum_prep_params[16] = {0};
  unsigned sum_prep_lut_codes = 0, sum_prep_tipmap = 0;
  bool sum_prep_has_lut = false;

  // --- device-resident data ---
  std::vector<double *> d_clv;        // [nodes], nullptr for coded tips
  std::vector<uint8_t *> d_codes;     // [tips], nullptr unless coded
  unsigned * d_scalers = nullptr;     // [nscalers][N]
  double * d_pmat = nullptr;          // [nmat][R][S][Sp]
  double * d_pfrag = nullptr;         // 20 states: [nmat][R][400] the same matrices as compact MFMA A fragments
  double * d_lut = nullptr;           // [nmat][R][lut_codes][S]   (coded tips only)
  unsigned long long result_seq = 0;  // sequence word of the mapped result buffer (finish_reduction)
  uint8_t * d_s61_votes = nullptr;
  uint8_t * d_s61_ttscale = nullptr;  // per folded cherry: [codes][codes] scaling decision per pair of tip codes
  size_t s61_ttscale_cap = 0;
  // last scaling decisions of the 61-state rate-parallel launches: per parent vector up to
  // three orientations (an inner node's slot is computed from any two of its three
  // neighbours), each double-buffered
  struct PredSlot { unsigned long long key = ~0ULL; uint8_t * buf[2] = {nullptr, nullptr}; unsigned cur = 0; unsigned long long used = 0; };
  std::vector<PredSlot> s61_pred;     // [3 * clv index + slot]
  unsigned long long s61_pred_clock = 0;    // [op in launch][R][site] scaling votes of rate-parallel S61 launches
  unsigned lut_codes = 0;             // row count the LUTs were built for
  bool lut_stale = false;             // tipmap grew since the LUTs were built
  std::vector<double> pmat_brlen;     // last branch length per matrix (NaN = never set)
  std::vector<std::vector<unsigned>> pmat_params; // params_indices used per matrix
  unsigned * d_weights = nullptr;     // [N] pattern weights
  int * d_invariant = nullptr;        // [N], nullptr until needed
  unsigned long long * d_tipmap = nullptr; // [256] code -> state mask

  // queued P-matrix requests (flush_pmatrices)
  std::vector<unsigned> pend_midx;
  std::vector<double> pend_t;
  std::vector<int> pend_pos;          // matrix index -> position in the queue, -1 if absent
  ParamIdxHost pend_params = {};

  // model block: rates[R] weights[R] pinv[nrm] freqs[nrm][Sp] evals[nrm][Sp]
  //              evecs[nrm][S*Sp] ievecs[nrm][S*Sp]
  double * d_model = nullptr;
  std::vector<double> model_shadow;
  bool pmatrix_burst = false;         // the last model check was made by a P-matrix request (sync_model, light)
  bool eigen_touched = true;          // an eigen-system was recomputed or loaded since the last check
  size_t off_rates = 0, off_weights = 0, off_pinv = 0, off_freqs = 0,
         off_evals = 0, off_evecs = 0, off_ievecs = 0, model_len = 0;

  // pattern weights / invariant shadows (callers may also poke these)
  std::vector<unsigned> weights_shadow;
  bool invariant_uploaded = false;
  const int * invariant_host_seen = nullptr;
  unsigned tipmap_codes_uploaded = 0;

  // reductions
  double * d_partials = nullptr;      // [REDUCE_QUANTITIES * REDUCE_BLOCKS]
  unsigned * d_counter = nullptr;     // arrival tickets of the grid-wide reductions
  double * h_result = nullptr;        // pinned + device-mapped: final sums land here
  double * d_result = nullptr;        // device pointer of h_result
  bool fused_finish = true;           // the reduction kernels finish the sum themselves (no k_final_sum launch)
  // where the totals of the NEXT reduction launch go (set by the entry point before the launch):
  // the mapped result buffer + its sequence word, or a slot of a deferred result group
  struct Sink { double * dst = nullptr; unsigned long long * flag = nullptr; unsigned long long seq = 0; unsigned nq = 0; } sink;
  double * d_persite = nullptr;       // [N], allocated on first per-site request
  // device-resident Newton-Raphson (pllhip_newton_branch): control block in device memory, results + iterate
  // trail in mapped host memory (allocated on first use)
  void * d_newton = nullptr;
  double * h_newton = nullptr;        // pinned + mapped: [0..7] results, [8..103] trail, [112] sequence word
  double * hd_newton = nullptr;
  unsigned long long newton_seq = 0;
  int newton_capacity = -1;           // co-resident workgroups of the loop kernel (-1: not asked yet)
  int newton_resident = 0;            // blocks per wave that k_newton_mfma_resident keeps in registers (0: the streaming loop)
  const void * newton_fn = nullptr;
  size_t newton_lds = 0;
  // ... and the streaming form of the same loop (several partitions under one branch length share the chip)
  const void * newton_stream_fn = nullptr;
  size_t newton_stream_lds = 0;
  int newton_stream_capacity = 0;
  hipEvent_t newton_ready = nullptr;  // the control block of a multi-partition loop is initialised
  hipEvent_t newton_done = nullptr;   // this partition's instance of the last multi-partition loop has left the device

  // caller-keyed device sumtables (pointer value is the key)
  std::list<std::pair<const void *, double *>> sumtables;

  // host-mirror dirtiness
  bool pmat_host_dirty = false;

  pllhip_counters_t counters = {};

  DevicePlan plan;                    // schedule of the last whole-traversal launch
  BatchPlan * batch = nullptr;        // leader of a batch: the merged schedule of the last batched call

  // optional timing of the partials launches with HIP events on `stream`
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;  // pool, reused
  size_t prof_used = 0;
  double prof_bytes = 0.0;        // algorithmic bytes of the recorded launches
  double prof_min_bytes = 0.0;    // bytes those launches have to move (carried children not read)
  double prof_flops = 0.0;        // algorithmic flops of the recorded launches
  unsigned long long prof_ops = 0;
};

inline Engine * engine_of(const pll_partition_t * p) { return static_cast<Engine *>(p->engine); }
inline bool is_router(const pll_partition_t * p) { return !engine_of(p)->shards.empty(); }
// the engine that executes for `p`: its own, or the first shard's
inline Engine * exec_engine(const pll_partition_t * p)
{
  Engine * e = engine_of(p);
  return e->shards.empty() ? e : engine_of(e->shards[0]);
}

// --- host side model code (pll_model.cpp) ---
int update_eigen_host(pll_partition_t * p, unsigned params_index);
int eigen_decompose(unsigned S, unsigned Sp, const double * ex, const double * pi,
                    double * evecs, double * ievecs, double * evals);

// --- engine services (pll_core.hip) ---
double loglikelihood_impl(pll_partition_t * p, unsigned pc, int psc, unsigned cc, int csc,
                          int matrix_index, const unsigned * freqs_indices,
                          double * persite_lnl, const Engine::Sink * deferred);
int derivatives_impl(pll_partition_t * p, int parent_scaler_index, int child_scaler_index,
                     const double * brlens, unsigned count, const unsigned * params_indices,
                     const double * sumtable, const Engine::Sink * deferred,
                     double * out_df, double * out_ddf);
// timeout_s > 0: give up after that many seconds (PLL_ERROR_HIP_TIMEOUT; results that depend on another rank)
int wait_sequence(hipStream_t stream, const volatile unsigned long long * flag, unsigned long long seq,
                  double timeout_s = 0.0);
Engine * engine_create(pll_partition_t * p);
void engine_destroy(Engine * e);
int sync_model(pll_partition_t * p, bool light = false);              // host model arrays -> HBM if changed
int upload_tip_codes(pll_partition_t * p, unsigned tip);
int upload_tip_clv(pll_partition_t * p, unsigned tip, const double * host_clv);
int upload_tip_classes(pll_partition_t * p, unsigned tip, const unsigned * site_class, const unsigned long long * masks,
                       unsigned nclasses);
int upload_weights(pll_partition_t * p);
void invalidate_luts(pll_partition_t * p);

} // namespace pllhip
