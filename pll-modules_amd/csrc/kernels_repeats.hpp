// kernels_repeats.hpp -- site repeats, first step: cherries (PLL_ATTRIB_SITE_REPEATS).
//
// libpll-2's site repeats compute the vector of a node once per CLASS of sites that share the pattern of the
// subtree below it (the reference selects the attribute in its test harness, test/src/common.c:31, and carries
// the per-node class tables through its checkpoints, src/binary/binary_io_operations.c:231-282).  For a cherry --
// a tip x tip operation -- the class of a site is simply its pair of tip codes: at most codes^2 classes (a few
// hundred) however long the alignment is.  With the attribute set the 20-state family (and, with scalar kernels of
// the same structure at the end of this file, the 4-state family) therefore
//   * computes a cherry's vector per code pair (k_cherry_build: the arithmetic of the tip x tip operation on a
//     "pseudo alignment" whose sites are the pairs; a few hundred KiB that stay in the caches) and writes per site
//     only the class code (4 B instead of 640 B; the scaler counts live per class as well and are written out per
//     site, k_class_scaler_expand, only for a reader that needs them);
//   * hands the cherry to the operation above it as a "wide tip": a child read through a lookup table with one row
//     per class, [rate][class][20] = P . vector(class), built per traversal on the matrix cores with the very MFMA
//     sequence an inner child takes (k_pair_lut) -- so the operation's result is, bit for bit, what it computes
//     from the expanded vector;
//   * expands the vector to the site-indexed form only for a reader that needs it (k_cherry_expand: an edge lnL
//     or a sumtable AT the cherry, pllhip_get_clv, a checkpoint).
// What a caller can observe -- vectors, scaler counts, likelihoods, derivatives -- is identical to the attribute
// being off (tests/test_site_repeats.py).
// Round 4: the 2 .. 32-state family as well (kernels of the same structure at the end of this file); and without
// PLL_ATTRIB_PATTERN_TIP -- the combination libpll itself allows -- a tip set through pll_set_tip_states is a class node
// from the start (class = state mask; pll_core.hip, upload_tip_classes), read as a wide tip whose rows sit in LDS when
// they are few, also when the attribute is off.
//
// Second step: classes of whole subtrees.  A node both of whose children are known per class (tips, cherries, or
// nodes of this kind) is known per class itself: the class of a site is the pair (class below child 1, class below
// child 2).  The pairs that occur are numbered densely on the device once per topology (k_class_mark, a prefix sum,
// k_class_assign: what libpll's pll_update_repeats does with its lookup table; when the table of possible pairs would
// be too large, a hash table of the pairs that occur: k_class_hash_insert / _assign), and the node's table is rows1[class of child 1] * rows2[class of child 2], where
// the rows of a tip are its lookup table and the rows of a class node are P . its table (k_pair_lut again).  The
// frontier -- the first operation above that is computed per site -- reads its class children as wide tips.
#pragma once

#include "kernels_common.hpp"
#include "kernels_s20.hpp"
#include "kernels_s16.hpp"
#include "engine.h"

namespace pllhip {

// one operation of a traversal that is computed per class (device memory, part of the resident schedule)
struct CherryJob
{
  const double * lut1, * lut2;       // row tables of the two children, [rate][rows][states]: a tip's lookup table, or P . table of a class node
  const unsigned * rep;              // [2 * classes]: the children's classes of each class; null = a cherry (class = code1 * codes + code2)
  unsigned rows1, rows2;             // rows per rate of the two tables
  unsigned nclasses;
  double * table;                    // blocked pseudo-CLV [class block][rate][unit] (4 states: [class][rate][4])
  uint8_t * flags;                   // per class: the vector was scaled
  unsigned * counts;                 // per class: scaler count (own decision + the children's counts), or null: the operation has no scale buffer
  const unsigned * cnt1, * cnt2;     // per-class counts of the children (class nodes with a scale buffer), or null
};

// one row table of a traversal: the table of a class node seen through the P-matrix of the branch above it
struct PairLutJob
{
  const double * table;
  const double * pfrag;              // compact A fragments [rate][400] of the branch's matrices (4 states: the matrices)
  double * out;                      // [rate][rows][states]
  unsigned nrows;
};

// the table of a class operation.  grid = (class blocks of the largest job / 4, jobs), block = 256 (a wave per
// 32-class block)
template <unsigned RT>
__global__ __launch_bounds__(256) void k_cherry_build(const CherryJob * jobs, unsigned ncodes)
{
  const CherryJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned npairs = job.nclasses, npblk = (npairs + S20_BS - 1) / S20_BS;
  const unsigned blk = blockIdx.x * 4 + wave;
  if (blk >= npblk) return;
  const double * lut1 = as_global(job.lut1), * lut2 = as_global(job.lut2);
  const unsigned * rep = as_global(job.rep);
  double * table = as_global(job.table);
  uint8_t * flags = as_global(job.flags);
  const unsigned pe = blk * S20_BS + 2 * n, po = pe + 1;
  // classes beyond the table repeat class 0: their columns are never read
  const unsigned ce = pe < npairs ? pe : 0, co = po < npairs ? po : 0;
  const unsigned ae = rep ? rep[2 * ce] : ce / ncodes, be = rep ? rep[2 * ce + 1] : ce % ncodes;
  const unsigned ao = rep ? rep[2 * co] : co / ncodes, bo = rep ? rep[2 * co + 1] : co % ncodes;
  double2 X[RT][5];
  int small_e = 1, small_o = 1;
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
    double2 t2[5];
    s20_child_tip(lut1 + (size_t)r * job.rows1 * 20, ae, ao, q, X[r]);
    s20_child_tip(lut2 + (size_t)r * job.rows2 * 20, be, bo, q, t2);
#pragma unroll
    for (int k = 0; k < 5; ++k)
    {
      X[r][k].x *= t2[k].x;
      X[r][k].y *= t2[k].y;
      small_e &= (X[r][k].x < SCALE_THRESHOLD);
      small_o &= (X[r][k].y < SCALE_THRESHOLD);
    }
  }
  double fe = 1.0, fo = 1.0;
  if (job.counts)
  {
    small_e = s20_and_q(small_e);
    small_o = s20_and_q(small_o);
    fe = small_e ? SCALE_FACTOR : 1.0;
    fo = small_o ? SCALE_FACTOR : 1.0;
    if (q == 0)
    {
      // the scaler count of the class: its own decision plus the counts of the children's classes
      const unsigned * cnt1 = as_global(job.cnt1), * cnt2 = as_global(job.cnt2);
      unsigned * counts = as_global(job.counts);
      if (pe < npairs) { flags[pe] = (uint8_t)small_e; counts[pe] = (unsigned)small_e + (cnt1 ? cnt1[ae] : 0u) + (cnt2 ? cnt2[be] : 0u); }
      if (po < npairs) { flags[po] = (uint8_t)small_o; counts[po] = (unsigned)small_o + (cnt1 ? cnt1[ao] : 0u) + (cnt2 ? cnt2[bo] : 0u); }
    }
  }
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
#pragma unroll
    for (int k = 0; k < 5; ++k) { X[r][k].x *= fe; X[r][k].y *= fo; }
    s20_store_d(table + ((size_t)blk * RT + r) * S20_UNIT, lane, X[r]);
  }
}

// the per-site scaler counts of a class node, for a reader that needs them (the counts live per class, like the
// vector: k_cherry_build).  grid = chunks, block = 256
__global__ __launch_bounds__(256) void k_class_scaler_expand(const unsigned * counts, const unsigned * pair, unsigned nalloc,
                                                             unsigned * ps)
{
  for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < nalloc; s += gridDim.x * 256u) ps[s] = counts[pair[s]];
}

// ---------------------------------------------------------------------------
// class maps (once per topology and orientation of a node; cached on the host side)
// ---------------------------------------------------------------------------
// classes of the two children per site: a tip's codes (bytes) or a class node's classes (32 bit)
struct ClassMapArgs
{
  const uint8_t * codes1, * codes2;          // tip children
  const unsigned * cls1, * cls2;             // class-node children
  unsigned n1, n2;                           // classes of the children
};

__device__ inline unsigned class_key(const ClassMapArgs & a, unsigned s)
{
  unsigned k1 = a.codes1 ? a.codes1[s] : a.cls1[s], k2 = a.codes2 ? a.codes2[s] : a.cls2[s];
  if (k1 >= a.n1) k1 = 0;
  if (k2 >= a.n2) k2 = 0;
  return k1 * a.n2 + k2;
}

// a cherry: every code pair is a class.  grid = chunks, block = 256
__global__ __launch_bounds__(256) void k_class_cherry(ClassMapArgs a, unsigned nalloc, unsigned * pair)
{
  for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < nalloc; s += gridDim.x * 256u)
    pair[s] = class_key(a, s);                            // (padding sites carry code 0)
}

// which pairs occur.  grid = chunks, block = 256; `seen` zeroed [n1 * n2]
__global__ __launch_bounds__(256) void k_class_mark(ClassMapArgs a, unsigned N, unsigned * seen)
{
  for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < N; s += gridDim.x * 256u) seen[class_key(a, s)] = 1u;
}

// exclusive prefix sum of `seen` (n <= 1024 * 4096), three small launches: sums of 4096-entry tiles, their prefix
// in one workgroup, then the entries
__global__ __launch_bounds__(1024) void k_class_scan_tiles(const unsigned * seen, unsigned n, unsigned * tile_sum)
{
  __shared__ unsigned part[16];
  unsigned v = 0;
  const unsigned base = blockIdx.x * 4096u + threadIdx.x * 4u;
  for (unsigned u = 0; u < 4; ++u) if (base + u < n) v += seen[base + u];
  for (int off = 32; off; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned t = 0; for (int w = 0; w < 16; ++w) t += part[w]; tile_sum[blockIdx.x] = t; }
}

__global__ __launch_bounds__(1024) void k_class_scan_top(unsigned * tile_sum, unsigned ntiles, unsigned * total)
{
  __shared__ unsigned buf[1024];
  const unsigned v = threadIdx.x < ntiles ? tile_sum[threadIdx.x] : 0u;
  buf[threadIdx.x] = v;
  __syncthreads();
  for (unsigned off = 1; off < 1024; off <<= 1)
  {
    const unsigned add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0u;
    __syncthreads();
    buf[threadIdx.x] += add;
    __syncthreads();
  }
  if (threadIdx.x < ntiles) tile_sum[threadIdx.x] = buf[threadIdx.x] - v;       // exclusive
  if (threadIdx.x == 1023) *total = buf[1023];
}

// More pairs of child classes than a table can hold: the pairs that occur go through a hash table (open addressing,
// `slots` a power of two >= 2 N; key = pair + 1, 0 = empty).  Which slot a pair lands in depends on the order of the
// inserts, so the numbering of the classes differs from run to run -- the classes themselves, and everything computed
// from them, do not.  grid = chunks, block = 256
__device__ inline unsigned class_hash(unsigned long long key, unsigned mask)
{
  key ^= key >> 33; key *= 0xff51afd7ed558ccdULL; key ^= key >> 33; key *= 0xc4ceb9fe1a85ec53ULL; key ^= key >> 33;
  return (unsigned)key & mask;
}

__device__ inline unsigned long long class_key64(const ClassMapArgs & a, unsigned s)
{
  unsigned k1 = a.codes1 ? a.codes1[s] : a.cls1[s], k2 = a.codes2 ? a.codes2[s] : a.cls2[s];
  if (k1 >= a.n1) k1 = 0;
  if (k2 >= a.n2) k2 = 0;
  return (unsigned long long)k1 * a.n2 + k2 + 1ULL;
}

__global__ __launch_bounds__(256) void k_class_hash_insert(ClassMapArgs a, unsigned N, unsigned long long * keys, unsigned * seen,
                                                           unsigned mask)
{
  for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < N; s += gridDim.x * 256u)
  {
    const unsigned long long key = class_key64(a, s);
    for (unsigned h = class_hash(key, mask); ; h = (h + 1) & mask)
    {
      const unsigned long long old = atomicCAS(keys + h, 0ULL, key);
      if (old == 0ULL) { seen[h] = 1u; break; }
      if (old == key) break;
    }
  }
}

// ... after the prefix sum over `seen` (which then holds the class number of every occupied slot)
__global__ __launch_bounds__(256) void k_class_hash_assign(ClassMapArgs a, unsigned N, unsigned nalloc, const unsigned long long * keys,
                                                           const unsigned * ids, unsigned mask, unsigned * pair)
{
  for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < nalloc; s += gridDim.x * 256u)
  {
    unsigned id = 0;
    if (s < N)
    {
      const unsigned long long key = class_key64(a, s);
      unsigned h = class_hash(key, mask);
      while (keys[h] != key) h = (h + 1) & mask;
      id = ids[h];
    }
    pair[s] = id;
  }
}

// seen[k] becomes the class number of pair k (for pairs that occur); the children's classes of every class.
// keys == nullptr: k IS the pair (table of possible pairs); else keys[k] - 1 is (hash table)
__global__ __launch_bounds__(1024) void k_class_scan_apply(unsigned * seen, unsigned n, const unsigned * tile_sum,
                                                           unsigned n2, unsigned * rep, unsigned rep_cap,
                                                           const unsigned long long * keys)
{
  __shared__ unsigned part[16];
  const unsigned base = blockIdx.x * 4096u + threadIdx.x * 4u;
  unsigned f[4], v = 0;
  for (unsigned u = 0; u < 4; ++u) { f[u] = (base + u < n) ? seen[base + u] : 0u; v += f[u]; }
  // exclusive prefix of v over the workgroup
  unsigned incl = v;
  for (int off = 1; off < 64; off <<= 1) { const unsigned t = __shfl_up(incl, off, 64); if ((threadIdx.x & 63) >= (unsigned)off) incl += t; }
  if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = incl;
  __syncthreads();
  unsigned before = tile_sum[blockIdx.x];
  for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) before += part[w];
  unsigned id = before + incl - v;
  for (unsigned u = 0; u < 4; ++u)
    if (base + u < n)
    {
      seen[base + u] = id;
      if (f[u] && id < rep_cap)
      {
        const unsigned long long pr = keys ? keys[base + u] - 1ULL : (unsigned long long)(base + u);
        rep[2 * id] = (unsigned)(pr / n2);
        rep[2 * id + 1] = (unsigned)(pr % n2);
      }
      id += f[u];
    }
}

// the class of every site.  grid = chunks, block = 256
__global__ __launch_bounds__(256) void k_class_assign(ClassMapArgs a, unsigned N, unsigned nalloc, const unsigned * ids,
                                                      unsigned * pair)
{
  for (unsigned s = blockIdx.x * 256u + threadIdx.x; s < nalloc; s += gridDim.x * 256u)
    pair[s] = s < N ? ids[class_key(a, s)] : 0u;
}

// lookup tables of the wide tips of a traversal.  grid = (pair blocks / 4, jobs), block = 256,
// dynamic LDS = RT * 400 doubles (the compact fragments of the branch's matrices)
template <unsigned RT>
__global__ __launch_bounds__(256) void k_pair_lut(const PairLutJob * jobs)
{
  extern __shared__ double cfrag[];
  const PairLutJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned npairs = job.nrows;
  const double * pf = as_global(job.pfrag);
  staged_copy<8>(cfrag, pf, RT * S20_CFRAGS);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned npblk = (npairs + S20_BS - 1) / S20_BS;
  const unsigned blk = blockIdx.x * 4 + wave;
  if (blk >= npblk) return;
  const double * table = as_global(job.table);
  double * out = as_global(job.out);
  const unsigned pe = blk * S20_BS + 2 * n, po = pe + 1;
#pragma unroll
  for (unsigned r = 0; r < RT; ++r)
  {
    double2 t[5];
    s20_child_inner_c(table + ((size_t)blk * RT + r) * S20_UNIT, cfrag + r * S20_CFRAGS, lane, t, false);
#pragma unroll
    for (unsigned k = 0; k < 5; ++k)
    {
      const unsigned i = s20_row(k, q);
      if (pe < npairs) out[((size_t)r * npairs + pe) * 20 + i] = t[k].x;
      if (po < npairs) out[((size_t)r * npairs + po) * 20 + i] = t[k].y;
    }
  }
}

// the site-indexed vector of a cherry, for a reader that needs it.  grid = site blocks / 4, block = 256
__global__ __launch_bounds__(256) void k_cherry_expand(const double * table, const unsigned * pair,
                                                       unsigned nblk, unsigned R, double * clv)
{
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  for (unsigned blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    const unsigned pe = pair[site0], po = pair[site0 + 1];
    for (unsigned r = 0; r < R; ++r)
    {
      const double * ue = table + ((size_t)(pe / S20_BS) * R + r) * S20_UNIT + (q * 16 + (pe % S20_BS) / 2) * 2 + (pe & 1u);
      const double * uo = table + ((size_t)(po / S20_BS) * R + r) * S20_UNIT + (q * 16 + (po % S20_BS) / 2) * 2 + (po & 1u);
      double2 t[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) t[k] = make_double2(ue[k * 128], uo[k * 128]);
      s20_store_d(clv + ((size_t)blk * R + r) * S20_UNIT, lane, t);
    }
  }
}

// ---------------------------------------------------------------------------
// the same for the 4-state family (kernels_s4.hpp; vectors in the API layout [site][rate][4], tip tables
// [rate][16 codes][4]): at most 256 classes per cherry
// ---------------------------------------------------------------------------
// grid = (ceil(pairs / 256), jobs), block = 256: a thread per class
__global__ __launch_bounds__(256) void k_cherry_build_s4(const CherryJob * jobs, unsigned R, unsigned ncodes)
{
  const CherryJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned npairs = job.nclasses;
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= npairs) return;
  const double * lut1 = as_global(job.lut1), * lut2 = as_global(job.lut2);
  const unsigned * rep = as_global(job.rep);
  double * table = as_global(job.table);
  const unsigned a = rep ? rep[2 * p] : p / ncodes, b = rep ? rep[2 * p + 1] : p % ncodes;
  bool small = true;
  for (unsigned r = 0; r < R; ++r)
    for (unsigned i = 0; i < 4; ++i)
    {
      const double v = lut1[((size_t)r * job.rows1 + a) * 4 + i] * lut2[((size_t)r * job.rows2 + b) * 4 + i];
      table[((size_t)p * R + r) * 4 + i] = v;
      small = small && (v < SCALE_THRESHOLD);
    }
  if (job.counts)
  {
    const unsigned * cnt1 = as_global(job.cnt1), * cnt2 = as_global(job.cnt2);
    as_global(job.flags)[p] = small ? 1 : 0;
    as_global(job.counts)[p] = (small ? 1u : 0u) + (cnt1 ? cnt1[a] : 0u) + (cnt2 ? cnt2[b] : 0u);
    if (small)
      for (unsigned e = 0; e < R * 4; ++e) table[(size_t)p * R * 4 + e] *= SCALE_FACTOR;
  }
}

// [rate][class][4] = P(rate) . vector(class), with the association of s4_half_matvec:
// (P_i0 x_0 + P_i1 x_1) + (P_i2 x_2 + P_i3 x_3).  grid = (ceil(pairs * R / 256), jobs), block = 256;
// job.pfrag holds the branch's matrices [rate][4][4]
__global__ __launch_bounds__(256) void k_pair_lut_s4(const PairLutJob * jobs, unsigned R)
{
  const PairLutJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned npairs = job.nrows;
  const unsigned x = blockIdx.x * 256u + threadIdx.x;
  if (x >= npairs * R) return;
  const unsigned p = x / R, r = x % R;
  const double * P = as_global(job.pfrag) + r * 16;
  const double * v = as_global(job.table) + ((size_t)p * R + r) * 4;
  double * out = as_global(job.out) + ((size_t)r * npairs + p) * 4;
  for (unsigned i = 0; i < 4; ++i)
  {
    const double lo = P[i * 4 + 0] * v[0] + P[i * 4 + 1] * v[1];
    const double hi = P[i * 4 + 2] * v[2] + P[i * 4 + 3] * v[3];
    out[i] = (i < 2) ? lo + hi : hi + lo;
  }
}

// the site-indexed vector of a cherry: grid-stride over (site, rate) columns
__global__ __launch_bounds__(256) void k_cherry_expand_s4(const double * table, const unsigned * pair,
                                                          unsigned N, unsigned R, double * clv)
{
  const unsigned long long total = (unsigned long long)N * R;
  for (unsigned long long c = (unsigned long long)blockIdx.x * 256u + threadIdx.x; c < total;
       c += (unsigned long long)gridDim.x * 256u)
  {
    const unsigned long long n = c / R;
    const unsigned r = (unsigned)(c % R);
    const double * src = table + ((size_t)pair[n] * R + r) * 4;
    double * dst = clv + c * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
  }
}

// ---------------------------------------------------------------------------
// the same for the 2 .. 32-state family (kernels_s16.hpp; KS = ceil(S / 4) k-steps, units of 4 KS rows, tip tables
// [rate][code][S]) -- the reference runs its 5-state branch-length test under the attribute (test/runtest.py:45-51,
// test/src/common.c:31).  The rate count is a run-time value here: a class block is stored unscaled rate by rate and
// brought up by 2^256 afterwards where the vote over all rates says so -- k_partials_s16's rule, hence its bits.
// ---------------------------------------------------------------------------
// grid = (class blocks of the largest job / 4, jobs), block = 256
template <unsigned KS>
__global__ __launch_bounds__(256) void k_cherry_build_s16(const CherryJob * jobs, unsigned ncodes, unsigned R, unsigned S)
{
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const CherryJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned npairs = job.nclasses, npblk = (npairs + S20_BS - 1) / S20_BS;
  const unsigned blk = blockIdx.x * 4 + wave;
  if (blk >= npblk) return;
  const double * lut1 = as_global(job.lut1), * lut2 = as_global(job.lut2);
  const unsigned * rep = as_global(job.rep);
  double * table = as_global(job.table);
  uint8_t * flags = as_global(job.flags);
  const unsigned pe = blk * S20_BS + 2 * n, po = pe + 1;
  const unsigned ce = pe < npairs ? pe : 0, co = po < npairs ? po : 0;
  const unsigned ae = rep ? rep[2 * ce] : ce / ncodes, be = rep ? rep[2 * ce + 1] : ce % ncodes;
  const unsigned ao = rep ? rep[2 * co] : co / ncodes, bo = rep ? rep[2 * co + 1] : co % ncodes;
  int small_e = 1, small_o = 1;
  for (unsigned r = 0; r < R; ++r)
  {
    double2 t1[KS], t2[KS];
    s16_child_tip<KS>(lut1 + (size_t)r * job.rows1 * S, ae, ao, q, S, t1);
    s16_child_tip<KS>(lut2 + (size_t)r * job.rows2 * S, be, bo, q, S, t2);
#pragma unroll
    for (unsigned v = 0; v < KS; ++v)
    {
      t1[v].x *= t2[v].x;
      t1[v].y *= t2[v].y;
      small_e &= (t1[v].x < SCALE_THRESHOLD);      // rows >= S are zero: they never veto
      small_o &= (t1[v].y < SCALE_THRESHOLD);
    }
    s16_store_d<KS>(table + ((size_t)blk * R + r) * UNIT, lane, t1);
  }
  if (!job.counts) return;
  small_e = s20_and_q(small_e);
  small_o = s20_and_q(small_o);
  if (__any(small_e | small_o))
  {
    const double fe = small_e ? SCALE_FACTOR : 1.0, fo = small_o ? SCALE_FACTOR : 1.0;
    for (unsigned r = 0; r < R; ++r)
    {
      double * unit = table + ((size_t)blk * R + r) * UNIT;
      double2 t[KS];
      s16_load_d<KS>(unit, lane, t);
#pragma unroll
      for (unsigned v = 0; v < KS; ++v) { t[v].x *= fe; t[v].y *= fo; }
      s16_store_d<KS>(unit, lane, t);
    }
  }
  if (q == 0)
  {
    const unsigned * cnt1 = as_global(job.cnt1), * cnt2 = as_global(job.cnt2);
    unsigned * counts = as_global(job.counts);
    if (pe < npairs) { flags[pe] = (uint8_t)small_e; counts[pe] = (unsigned)small_e + (cnt1 ? cnt1[ae] : 0u) + (cnt2 ? cnt2[be] : 0u); }
    if (po < npairs) { flags[po] = (uint8_t)small_o; counts[po] = (unsigned)small_o + (cnt1 ? cnt1[ao] : 0u) + (cnt2 ? cnt2[bo] : 0u); }
  }
}

// row tables [rate][rows][S] = P . table(class), with the MFMA sequence an inner child takes (s16_child_inner).
// job.pfrag: the branch's matrices [rate][S][Sp].  grid = (row blocks / 4, jobs), block = 256,
// dynamic LDS = R * s16_fr(KS) doubles
template <unsigned KS>
__global__ __launch_bounds__(256) void k_pair_lut_s16(const PairLutJob * jobs, unsigned R, unsigned S, unsigned Sp)
{
  extern __shared__ double frag[];
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const PairLutJob job = plan_fetch(jobs + blockIdx.y);
  const unsigned npairs = job.nrows;
  s16_fill_frags<KS>(frag, as_global(job.pfrag), R, S, Sp);
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  const unsigned npblk = (npairs + S20_BS - 1) / S20_BS;
  const unsigned blk = blockIdx.x * 4 + wave;
  if (blk >= npblk) return;
  const double * table = as_global(job.table);
  double * out = as_global(job.out);
  const unsigned pe = blk * S20_BS + 2 * n, po = pe + 1;
  for (unsigned r = 0; r < R; ++r)
  {
    double2 t[KS];
    s16_child_inner<KS>(table + ((size_t)blk * R + r) * UNIT, frag + r * s16_fr(KS), lane, t);
#pragma unroll
    for (unsigned v = 0; v < KS; ++v)
    {
      const unsigned i = 4 * v + q;
      if (i >= S) continue;
      if (pe < npairs) out[((size_t)r * npairs + pe) * S + i] = t[v].x;
      if (po < npairs) out[((size_t)r * npairs + po) * S + i] = t[v].y;
    }
  }
}

// the site-indexed vector of a class node.  grid = site blocks / 4, block = 256
template <unsigned KS>
__global__ __launch_bounds__(256) void k_cherry_expand_s16(const double * table, const unsigned * pair,
                                                           unsigned nblk, unsigned R, double * clv)
{
  constexpr unsigned UNIT = 4 * KS * S20_BS;
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned q = lane >> 4, n = lane & 15;
  for (unsigned blk = blockIdx.x * 4 + wave; blk < nblk; blk += gridDim.x * 4)
  {
    const size_t site0 = (size_t)blk * S20_BS + 2 * n;
    const unsigned pe = pair[site0], po = pair[site0 + 1];
    for (unsigned r = 0; r < R; ++r)
    {
      const double * ue = table + ((size_t)(pe / S20_BS) * R + r) * UNIT + (q * 16 + (pe % S20_BS) / 2) * 2 + (pe & 1u);
      const double * uo = table + ((size_t)(po / S20_BS) * R + r) * UNIT + (q * 16 + (po % S20_BS) / 2) * 2 + (po & 1u);
      double2 t[KS];
#pragma unroll
      for (unsigned v = 0; v < KS; ++v) t[v] = make_double2(ue[v * 128], uo[v * 128]);
      s16_store_d<KS>(clv + ((size_t)blk * R + r) * UNIT, lane, t);
    }
  }
}

} // namespace pllhip
