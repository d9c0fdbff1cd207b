// merge.hip -- stage 2 (BWT of the parse, inverted lists) and stage 3 (final BWT / SA).
//
// Stage 2 replaces bwtparse.c:212-322; stage 3 replaces bwt() (pfbwt.cpp:109-242), its
// writers fwrite_chars_same_suffix{,_sa,_ssa} (pfbwt.cpp:520-676) and the threaded variant
// (pfthreads.hpp:83-518).  The reference walks SA(D) serially and fputc()s each char; here
// the walk is a data-parallel decomposition (the same one pfthreads.hpp:456-493 uses for its
// thread ranges): every dictionary suffix longer than w contributes occ(word) output chars,
// an exclusive scan of those counts fixes every output offset, and then
//   * fill / full-word entries are expanded output-centrically (each thread owns 16
//     consecutive BWT bytes, finds its SA(D) slot by binary search and walks forward),
//   * groups of equal suffixes whose members disagree (or any multi-word group when SA values
//     are requested) are merged by rank computation over the members' inverted lists - the
//     data-parallel form of the reference's heap merge (pfbwt.cpp:537-556).
#include "kernels.hpp"
#include "prims.hpp"
#include "devutil.hpp"

namespace pfp {

static constexpr int TB = 256;

// ------------------------------------------------------------------ dictionary index

__global__ void pos_word_kernel(const uint8_t *__restrict__ b, uint64_t N, const uint32_t *__restrict__ inc,
                                uint32_t *__restrict__ pos_word, uint32_t *__restrict__ wend, uint32_t d) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  bool term = b[i] == kEndOfWord;
  uint32_t wd = inc[i] - (term ? 1u : 0u);
  pos_word[i] = wd;
  if (term) wend[wd] = (uint32_t)i;
  if (i == N - 1) wend[d] = (uint32_t)i;
}

void build_dict_index(pfp_ctx *c, const Dictionary &D, DictIndex &ix) {
  const uint64_t N = D.dsize;
  ix.pos_word.alloc(c, N);
  ix.wend.alloc(c, D.d + 1);
  DBuf<uint32_t> inc(c, N);
  inclusive_count_eq_u8(c, D.bytes.p, kEndOfWord, inc.p, N);
  hipLaunchKernelGGL(pos_word_kernel, dim3(cdiv(N, TB)), dim3(TB), 0, c->stream, D.bytes.p, N, inc.p, ix.pos_word.p,
                     ix.wend.p, (uint32_t)D.d);
  PFP_HIP(hipGetLastError());
}

// lexicographic rank of every word = number of full-word suffixes before it in SA(D)
__global__ void word_start_flags_kernel(const uint8_t *__restrict__ b, uint64_t N, const uint32_t *__restrict__ sa,
                                        uint32_t *__restrict__ flag) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  uint32_t i = sa[t];
  bool start = (i == 0 || b[i - 1] == kEndOfWord) && i != N - 1;
  flag[t] = start ? 1u : 0u;
}
__global__ void lexrank_scatter_kernel(uint64_t N, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ flag,
                                       const uint32_t *__restrict__ scan, const uint32_t *__restrict__ pos_word,
                                       uint32_t *__restrict__ lexrank) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N || !flag[t]) return;
  lexrank[pos_word[sa[t]]] = scan[t];
}

void compute_lexrank(pfp_ctx *c, const Dictionary &D, const SuffixOrder &so, DictIndex &ix) {
  const uint64_t N = D.dsize;
  ix.lexrank.alloc(c, D.d);
  DBuf<uint32_t> flag(c, N), scan(c, N);
  hipLaunchKernelGGL(word_start_flags_kernel, dim3(cdiv(N, TB)), dim3(TB), 0, c->stream, D.bytes.p, N, so.sa.p, flag.p);
  exclusive_sum_u32(c, flag.p, scan.p, N);
  hipLaunchKernelGGL(lexrank_scatter_kernel, dim3(cdiv(N, TB)), dim3(TB), 0, c->stream, N, so.sa.p, flag.p, scan.p,
                     ix.pos_word.p, ix.lexrank.p);
  PFP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ stage 2: BWT of the parse

// bwtparse.c:242-267: BWT(P)[j] = P[SA[j]-1]; bwlast = last of the phrase before that one
// (cyclically), bwsai = sai of that phrase; SA[j]==0 -> dummy zeros.
__global__ void parse_gather_kernel(uint64_t P, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ sym,
                                    const uint8_t *__restrict__ last, const uint64_t *__restrict__ sai,
                                    uint32_t *__restrict__ bwtp, uint8_t *__restrict__ bwlast,
                                    uint64_t *__restrict__ bwsai, uint32_t *__restrict__ jidx) {
  uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > P) return;
  uint64_t s = sa[j];
  jidx[j] = (uint32_t)j;
  if (s == 0) {
    bwtp[j] = 0; bwlast[j] = 0; if (bwsai) bwsai[j] = 0;
  } else {
    bwtp[j] = sym[s - 1];
    bwlast[j] = (s == 1) ? last[P - 1] : last[s - 2];
    if (bwsai) bwsai[j] = sai[s - 1];
  }
}

void parse_bwt(pfp_ctx *c, const uint32_t *parse_sym, uint64_t P, const uint8_t *last, const uint64_t *sai,
               const uint32_t *occ_lex, uint64_t d, ParseBWT &out) {
  (void)occ_lex;
  PFP_REQUIRE(P >= 2, PFP_ESHORT, "parse has fewer than 2 phrases (bwtparse.c:244)");
  out.P = P;
  DBuf<uint32_t> sym(c, P + 1);
  PFP_HIP(hipMemcpyAsync(sym.p, parse_sym, P * 4, hipMemcpyDeviceToDevice, c->stream));
  PFP_HIP(hipMemsetAsync(sym.p + P, 0, 4, c->stream));
  SuffixOrder so;
  sort_int_suffixes(c, sym.p, P + 1, so);
  if (c->debug) validate_int_sa(c, sym.p, so);
  out.rounds = so.rounds;
  out.ilist.alloc(c, P + 1);
  out.bwlast.alloc(c, P + 1);
  if (sai) out.bwsai.alloc(c, P + 1);
  DBuf<uint32_t> bwtp(c, P + 1), bwtp_s(c, P + 1), jidx(c, P + 1);
  hipLaunchKernelGGL(parse_gather_kernel, dim3(cdiv(P + 1, TB)), dim3(TB), 0, c->stream, P, so.sa.p, sym.p, last, sai,
                     bwtp.p, out.bwlast.p, sai ? out.bwsai.p : (uint64_t *)nullptr, jidx.p);
  // bwtparse.c:281-303: positions grouped by symbol, ascending inside a group == stable sort
  sort_pairs_u32_u32(c, bwtp.p, bwtp_s.p, jidx.p, out.ilist.p, P + 1, 0, bits_for(d));
  PFP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ stage 3: merge

// per SA(D) slot: output count and the char that precedes the suffix (1 = "full word")
__global__ void slot_info_kernel(const uint8_t *__restrict__ b, uint64_t N, uint32_t d, int w,
                                 const uint32_t *__restrict__ sa, const uint32_t *__restrict__ pos_word,
                                 const uint32_t *__restrict__ wend, const uint32_t *__restrict__ wocc,
                                 uint32_t *__restrict__ cnt, uint8_t *__restrict__ pc) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  uint32_t i = sa[t];
  uint32_t wd = pos_word[i];
  bool valid = wd < d && (wend[wd] - i) > (uint32_t)w;     // pfbwt.cpp:151
  cnt[t] = valid ? wocc[wd] : 0u;
  pc[t] = valid ? (i == 0 ? kEndOfWord : b[i - 1]) : 0;
}

__global__ void group_flags_kernel(uint64_t N, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ rank,
                                   const uint32_t *__restrict__ cnt, const uint8_t *__restrict__ pc,
                                   int any_multi_is_hard, uint8_t *__restrict__ hard) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N || cnt[t] == 0) return;
  uint32_t g = rank[sa[t]];
  if (g == t) return;
  if (any_multi_is_hard || pc[t] != pc[g]) hard[g] = 1;
}

enum : uint8_t { CLS_NONE = 0, CLS_FILL = 1, CLS_FULL = 2, CLS_HARD = 3 };
__global__ void classify_kernel(uint64_t N, const uint32_t *__restrict__ sa, const uint32_t *__restrict__ rank,
                                const uint32_t *__restrict__ cnt, const uint8_t *__restrict__ pc,
                                const uint8_t *__restrict__ hard, uint8_t *__restrict__ cls) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N) return;
  uint8_t k = CLS_NONE;
  if (cnt[t]) k = (pc[t] == kEndOfWord) ? CLS_FULL : (hard[rank[sa[t]]] ? CLS_HARD : CLS_FILL);
  cls[t] = k;
}

__global__ void wistart_kernel(uint32_t d, const uint32_t *__restrict__ lexrank, const uint32_t *__restrict__ istart_lex,
                               uint32_t *__restrict__ wistart) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d) wistart[j] = istart_lex[lexrank[j]] + 1;   // +1: ilist[0] is the EOS symbol (pfbwt.cpp:389)
}

struct MergeArgs {
  uint64_t N, n_out; uint32_t d; int w; int want_sa;
  const uint8_t *b; const uint32_t *sa, *rank, *pos_word, *wend, *wocc, *wistart;
  const uint32_t *cnt; const uint8_t *pc, *cls; const uint64_t *off;
  const uint32_t *ilist; const uint8_t *bwlast; const uint64_t *bwsai;
  uint8_t *bwt; uint64_t *out_sa;
};

__device__ __forceinline__ uint8_t fix_char(uint8_t ch) { return ch == kDollar ? 0 : ch; }  // pfbwt.cpp:126

// largest t in [0,N) with off[t] <= x  (off has N+1 entries, off[N] = total > x)
__device__ __forceinline__ uint64_t find_slot(const uint64_t *__restrict__ off, uint64_t N, uint64_t x) {
  uint64_t lo = 0, hi = N;          // invariant: off[lo] <= x < off[hi]
  while (hi - lo > 1) {
    uint64_t mid = (lo + hi) >> 1;
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

// output-centric expansion of fill and full-word entries: 16 BWT bytes per thread
__global__ __launch_bounds__(256) void expand_kernel(MergeArgs a) {
  uint64_t x0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
  if (x0 >= a.n_out) return;
  uint64_t t = find_slot(a.off, a.N, x0);
  uint32_t r[4] = {0, 0, 0, 0};
  uint64_t next_off = a.off[t + 1];
  uint64_t base = a.off[t];
  int nb = (a.n_out - x0) >= 16 ? 16 : (int)(a.n_out - x0);
  for (int k = 0; k < nb; k++) {
    uint64_t x = x0 + k;
    while (x >= next_off) { t++; base = next_off; next_off = a.off[t + 1]; }
    uint8_t cl = a.cls[t];
    uint8_t ch = 0;
    if (cl == CLS_FILL) {
      ch = fix_char(a.pc[t]);
      if (a.want_sa) {
        uint32_t i = a.sa[t];
        uint32_t wd = a.pos_word[i];
        uint64_t pos = a.ilist[a.wistart[wd] + (uint32_t)(x - base)];
        a.out_sa[x] = a.bwsai[pos] - (uint64_t)(a.wend[wd] - i);
      }
    } else if (cl == CLS_FULL) {
      uint32_t i = a.sa[t];
      uint32_t wd = a.pos_word[i];
      uint64_t pos = a.ilist[a.wistart[wd] + (uint32_t)(x - base)];
      ch = a.bwlast[pos];
      if (a.want_sa) a.out_sa[x] = (x == 0) ? a.n_out - 1 : a.bwsai[pos] - (uint64_t)(a.wend[wd] - i);
    }
    r[k >> 2] |= (uint32_t)ch << (8 * (k & 3));
  }
  if (nb == 16) *reinterpret_cast<uint4 *>(a.bwt + x0) = make_uint4(r[0], r[1], r[2], r[3]);
  else for (int k = 0; k < nb; k++) a.bwt[x0 + k] = (uint8_t)(r[k >> 2] >> (8 * (k & 3)));
}

// one wave per hard group: rank every (member, occurrence) by its BWT(P) position
__global__ __launch_bounds__(64) void hard_groups_kernel(MergeArgs a, const uint32_t *__restrict__ hlist, uint32_t nh,
                                                         unsigned long long *__restrict__ hard_chars) {
  uint32_t gi = blockIdx.x;
  if (gi >= nh) return;
  const uint64_t g = hlist[gi];
  const int lane = threadIdx.x;
  // member count: consecutive slots whose suffix has rank g
  uint32_t k = 0;
  for (;;) {
    uint64_t t = g + k + lane;
    bool in = t < a.N && a.rank[a.sa[t]] == (uint32_t)g;
    unsigned long long m = __ballot(in);
    if (m == ~0ULL) { k += 64; continue; }
    k += __ffsll((long long)~m) - 1;
    break;
  }
  const uint64_t base = a.off[g];
  const uint64_t C = a.off[g + k] - base;
  if (lane == 0) atomicAdd(hard_chars, (unsigned long long)C);
  for (uint64_t e = lane; e < C; e += 64) {
    // member holding element e
    uint32_t lo = 0, hi = k;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (a.off[g + mid] - base <= e) lo = mid; else hi = mid; }
    const uint32_t m = lo;
    const uint32_t j = (uint32_t)(e - (a.off[g + m] - base));
    const uint32_t i = a.sa[g + m];
    const uint32_t wd = a.pos_word[i];
    const uint32_t pos = a.ilist[a.wistart[wd] + j];
    uint64_t r = j;
    for (uint32_t m2 = 0; m2 < k; m2++) {
      if (m2 == m) continue;
      const uint32_t wd2 = a.pos_word[a.sa[g + m2]];
      const uint32_t *lst = a.ilist + a.wistart[wd2];
      uint32_t l2 = 0, h2 = a.wocc[wd2];         // # entries < pos
      while (l2 < h2) { uint32_t mid = (l2 + h2) >> 1; if (lst[mid] < pos) l2 = mid + 1; else h2 = mid; }
      r += l2;
    }
    const uint64_t x = base + r;
    a.bwt[x] = fix_char(a.pc[g + m]);
    if (a.want_sa) a.out_sa[x] = a.bwsai[pos] - (uint64_t)(a.wend[wd] - i);
  }
}

void merge_bwt(pfp_ctx *c, const Dictionary &D, const DictIndex &ix, const SuffixOrder &so, const ParseBWT &pb,
               const uint32_t *occ_lex, int w, int flags, uint64_t expect_n_out, BwtOutputs &out) {
  const uint64_t N = D.dsize;
  const uint32_t d = (uint32_t)D.d;
  PFP_REQUIRE(!flags || pb.bwsai.p, PFP_EINVAL, "SA output requested without sa info");
  // istart in lexicographic order (pfbwt.cpp:388-396), looked up per word
  DBuf<uint32_t> istart_lex(c, d), wistart(c, d);
  exclusive_sum_u32(c, occ_lex, istart_lex.p, d);
  hipLaunchKernelGGL(wistart_kernel, dim3(cdiv(d, TB)), dim3(TB), 0, c->stream, d, ix.lexrank.p, istart_lex.p, wistart.p);
  DBuf<uint32_t> cnt(c, N + 1);
  DBuf<uint8_t> pc(c, N), hard(c, N), cls(c, N);
  DBuf<uint64_t> off(c, N + 1);
  PFP_HIP(hipMemsetAsync(cnt.p + N, 0, 4, c->stream));
  hard.zero();
  hipLaunchKernelGGL(slot_info_kernel, dim3(cdiv(N, TB)), dim3(TB), 0, c->stream, D.bytes.p, N, d, w, so.sa.p,
                     ix.pos_word.p, ix.wend.p, D.wocc.p, cnt.p, pc.p);
  exclusive_sum_u32_u64(c, cnt.p, off.p, N + 1);
  hipLaunchKernelGGL(group_flags_kernel, dim3(cdiv(N, TB)), dim3(TB), 0, c->stream, N, so.sa.p, so.rank.p, cnt.p, pc.p,
                     flags ? 1 : 0, hard.p);
  hipLaunchKernelGGL(classify_kernel, dim3(cdiv(N, TB)), dim3(TB), 0, c->stream, N, so.sa.p, so.rank.p, cnt.p, pc.p,
                     hard.p, cls.p);
  DBuf<uint32_t> hlist(c, N), nh_d(c, 1);
  select_index_u32(c, hard.p, hlist.p, nh_d.p, N);
  PFP_HIP(hipMemcpyAsync(c->h_scalars, off.p + N, 8, hipMemcpyDeviceToHost, c->stream));
  PFP_HIP(hipMemcpyAsync(c->h_scalars + 1, nh_d.p, 4, hipMemcpyDeviceToHost, c->stream));
  sync(c);
  const uint64_t n_out = c->h_scalars[0];
  uint32_t nh; memcpy(&nh, c->h_scalars + 1, 4);
  PFP_REQUIRE(expect_n_out == 0 || n_out == expect_n_out, PFP_EFORMAT,
              "merge: sum of occurrence counts (" + std::to_string(n_out) + ") != text length + 1 (" +
                  std::to_string(expect_n_out) + ")");
  out.n_out = n_out;
  MergeArgs a{};
  a.N = N; a.n_out = n_out; a.d = d; a.w = w; a.want_sa = flags ? 1 : 0;
  a.b = D.bytes.p; a.sa = so.sa.p; a.rank = so.rank.p; a.pos_word = ix.pos_word.p; a.wend = ix.wend.p;
  a.wocc = D.wocc.p; a.wistart = wistart.p; a.cnt = cnt.p; a.pc = pc.p; a.cls = cls.p; a.off = off.p;
  a.ilist = pb.ilist.p; a.bwlast = pb.bwlast.p; a.bwsai = pb.bwsai.p; a.bwt = out.d_bwt; a.out_sa = out.d_sa;
  hipLaunchKernelGGL(expand_kernel, dim3(cdiv(cdiv64(n_out, 16), 256)), dim3(256), 0, c->stream, a);
  out.hard_groups = nh;
  if (nh) {
    DBuf<unsigned long long> hc(c, 1);
    hc.zero();
    hipLaunchKernelGGL(hard_groups_kernel, dim3(nh), dim3(64), 0, c->stream, a, hlist.p, nh, hc.p);
    out.hard_chars = read_scalar(c, hc.p);
  }
  PFP_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ output packing

// utils.c:112-129: low 5 bytes, little endian
__global__ void pack5_kernel(const uint64_t *__restrict__ v, uint64_t cnt, uint8_t *__restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  uint64_t x = v[i];
  uint8_t *o = out + 5 * i;
#pragma unroll
  for (int b = 0; b < 5; b++) o[b] = (uint8_t)(x >> (8 * b));
}
__global__ void unpack5_kernel(const uint8_t *__restrict__ in, uint64_t cnt, uint64_t *__restrict__ v) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  const uint8_t *p = in + 5 * i;
  uint64_t x = 0;
#pragma unroll
  for (int b = 0; b < 5; b++) x |= (uint64_t)p[b] << (8 * b);
  v[i] = x;
}
void pack5_dev(pfp_ctx *c, const uint64_t *vals, uint64_t cnt, uint8_t *out5) {
  if (!cnt) return;
  hipLaunchKernelGGL(pack5_kernel, dim3(cdiv(cnt, TB)), dim3(TB), 0, c->stream, vals, cnt, out5);
  PFP_HIP(hipGetLastError());
}
void unpack5_dev(pfp_ctx *c, const uint8_t *in5, uint64_t cnt, uint64_t *vals) {
  if (!cnt) return;
  hipLaunchKernelGGL(unpack5_kernel, dim3(cdiv(cnt, TB)), dim3(TB), 0, c->stream, in5, cnt, vals);
  PFP_HIP(hipGetLastError());
}

// run boundaries: .ssa = <j,SA[j]> for BWT[j] != BWT[j-1] incl. j=0 (pfbwt.cpp:169-174,184-189);
//                 .esa = <j,SA[j]> for BWT[j] != BWT[j+1] incl. j=n (pfbwt.cpp:175-179,225-229)
__global__ void run_flags_kernel(const uint8_t *__restrict__ bwt, uint64_t base, uint64_t cnt, uint64_t n_out,
                                 int run_end, uint8_t *__restrict__ flag) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  uint64_t x = base + i;
  bool f = run_end ? (x + 1 == n_out || bwt[x] != bwt[x + 1]) : (x == 0 || bwt[x] != bwt[x - 1]);
  flag[i] = f ? 1 : 0;
}
__global__ void write_pairs_kernel(const uint32_t *__restrict__ idx, uint32_t cnt, uint64_t base,
                                   const uint64_t *__restrict__ sa, uint8_t *__restrict__ out10) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cnt) return;
  uint64_t x = base + idx[i], v = sa[x];
  uint8_t *o = out10 + 10 * (uint64_t)i;
#pragma unroll
  for (int b = 0; b < 5; b++) { o[b] = (uint8_t)(x >> (8 * b)); o[5 + b] = (uint8_t)(v >> (8 * b)); }
}

uint64_t sample_runs_dev(pfp_ctx *c, const uint8_t *bwt, const uint64_t *sa, uint64_t n_out, bool run_end,
                         DBuf<uint8_t> &out10) {
  const uint64_t CH = 1ull << 30;
  std::vector<DBuf<uint8_t>> parts;
  std::vector<uint64_t> counts;
  uint64_t total = 0;
  for (uint64_t base = 0; base < n_out; base += CH) {
    uint64_t cnt = std::min(CH, n_out - base);
    DBuf<uint8_t> flag(c, cnt);
    DBuf<uint32_t> idx(c, cnt), nsel(c, 1);
    hipLaunchKernelGGL(run_flags_kernel, dim3(cdiv(cnt, TB)), dim3(TB), 0, c->stream, bwt, base, cnt, n_out,
                       run_end ? 1 : 0, flag.p);
    select_index_u32(c, flag.p, idx.p, nsel.p, cnt);
    uint32_t k = read_scalar(c, nsel.p);
    DBuf<uint8_t> part(c, (size_t)k * 10);
    if (k) hipLaunchKernelGGL(write_pairs_kernel, dim3(cdiv(k, TB)), dim3(TB), 0, c->stream, idx.p, k, base, sa, part.p);
    parts.push_back(std::move(part));
    counts.push_back(k);
    total += k;
  }
  out10.alloc(c, total * 10);
  uint64_t o = 0;
  for (size_t i = 0; i < parts.size(); i++) {
    if (counts[i]) PFP_HIP(hipMemcpyAsync(out10.p + o, parts[i].p, counts[i] * 10, hipMemcpyDeviceToDevice, c->stream));
    o += counts[i] * 10;
  }
  sync(c);
  PFP_HIP(hipGetLastError());
  return total;
}

}  // namespace pfp
