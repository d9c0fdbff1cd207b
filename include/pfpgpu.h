/* pfpgpu.h -- C ABI of libpfpgpu.so: MI355X (gfx950) prefix-free-parsing BWT builder.
 *
 * This is the drop-in boundary for the parse -> SA -> BWT hot path of alshai/Big-BWT.
 * The reference has no FFI; its stages are three executables glued by files, and the only
 * in-process ABI on the path is gsa/gsacak.h.  Each entry point below names the reference
 * interface it replaces (file:line under the reference tree).  Plain pointers and sizes only:
 * no C++/torch types cross this boundary.
 *
 * Conventions
 *   - every function returns PFP_OK (0) or a negative PFP_E* code; nothing calls exit()
 *     (the reference die()s: utils.c:12-16); pfp_last_error(ctx) gives a message.
 *   - "host" entry points take/return caller-owned host buffers in the reference's on-disk
 *     byte formats; "_dev" entry points take device pointers (hipMalloc'ed
 *     or torch CUDA tensors' data_ptr) and leave results in device memory.
 *   - one pfp_ctx per host thread / GPU; a ctx owns one HIP stream and a device memory pool.
 *   - all compute runs in HIP kernels on the ctx's device; there is no CPU fallback: if no
 *     GPU is usable pfp_ctx_create fails with PFP_ENODEV.
 */
#ifndef PFPGPU_H
#define PFPGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFP_OK 0
#define PFP_EINVAL (-1)     /* bad argument (w<4, p<10: newscan.cpp:537-544; -S with -s/-e: bigbwt:59-61) */
#define PFP_ENODEV (-2)     /* no usable HIP device */
#define PFP_EHIP (-3)       /* HIP runtime error */
#define PFP_ECOLLISION (-4) /* phrase-hash collision survived all reseeds (newscan.cpp:282-286) */
#define PFP_ELIMIT (-5)     /* size limit: parse > 2^32-2 words (bwtparse.c:93), dict >= 2^32-2 bytes */
#define PFP_EFORMAT (-6)    /* inconsistent stage inputs (pfbwt.cpp:498-512 style checks) */
#define PFP_ENOMEM (-7)
#define PFP_ESHORT (-8)     /* text shorter than the window / empty parse (bwtparse.c:244) */

/* output selection, same meaning as bigbwt -S / -s / -e (bigbwt:41-43) */
#define PFP_FLAG_SA 1
#define PFP_FLAG_SSA 2
#define PFP_FLAG_ESA 4

typedef struct pfp_ctx pfp_ctx;

/* number of HIP devices this process can see (0 without a GPU); `bigbwt -G 0` takes all of them */
int pfp_device_count(void);
int pfp_ctx_create(pfp_ctx **ctx, int device);
void pfp_ctx_destroy(pfp_ctx *ctx);
const char *pfp_last_error(const pfp_ctx *ctx);
const char *pfp_strerror(int code);
/* library / kernel build identification ("pfpgpu <ver> gfx950 ...") */
const char *pfp_version(void);
/* HIP stream the ctx launches on (hipStream_t), for callers that time with events */
void *pfp_ctx_stream(pfp_ctx *ctx);
void pfp_free(void *host_ptr);     /* frees host buffers returned by this library */
/* PFP_POOL_DEBUG=1 in the environment of pfp_ctx_create: every device block of the context gets an exact-size
 * allocation of its own with canary bands on both sides and a poison-filled body; bands are verified when a
 * block is released.  pfp_debug_check returns PFP_EHIP (message in pfp_last_error) once a band was damaged. */
int pfp_debug_check(pfp_ctx *ctx);
/* returns the context's cached (currently unused) device blocks to the driver */
void pfp_pool_trim(pfp_ctx *ctx);
/* out = {bytes held from the driver, peak of the bytes in use, bytes in use now, blocks handed out in debug mode};
 * "in use" counts what the holders asked for (a recycled block can be larger than the request: that shows in out[0]) */
int pfp_get_mem_stats(const pfp_ctx *ctx, uint64_t out[4]);
/* out = {allocations that reached the driver (hipMalloc) since the context was made, times a failed one made the
 * pool hand its cached blocks back}: a steady-state call adds nothing to either */
int pfp_get_pool_counters(const pfp_ctx *ctx, uint64_t out[2]);

/* ------------------------------------------------------------------------------------
 * Stage 1a: rolling Karp-Rabin window scan + phrase-boundary compaction.
 * Replaces KR_window::addchar + the trigger test of process_file (newscan.cpp:168-202,
 * 363-377; pscan.hpp:44-108).  ends[k] = text position of the last byte of phrase k, for
 * every trigger position (the final phrase, which ends in the w Dollars, is not listed).
 * *n_used = bytes actually parsed: parsing stops at the first byte <= 2 (newscan.cpp:364).
 * ------------------------------------------------------------------------------------ */
int pfp_scan(pfp_ctx *ctx, const uint8_t *text, uint64_t n, int w, uint64_t p,
             uint64_t **ends, uint64_t *n_ends, uint64_t *n_used);

/* ------------------------------------------------------------------------------------
 * Stage 1: the whole parser (newscanNT.x / pscan.x main: newscan.cpp:569-650).
 * Outputs are the reference's files as byte-exact buffers (library-allocated, pfp_free):
 *   dict  (.dict)  sorted phrases, each + 0x01, final 0x00      occ (.occ) u32[d]
 *   parse (.parse) u32[P] 1-based ranks                         last (.last) u8[P]
 *   sai   (.sai)   5-byte LE ints [P], only when want_sai
 * ------------------------------------------------------------------------------------ */
typedef struct {
  uint64_t n_used;
  uint8_t *dict;   uint64_t dict_size;
  uint32_t *occ;   uint64_t n_words;      /* d */
  uint32_t *parse; uint64_t n_phrases;    /* P */
  uint8_t *last;
  uint8_t *sai;                           /* 5*P bytes or NULL */
} pfp_parse_result;
int pfp_parse(pfp_ctx *ctx, const uint8_t *text, uint64_t n, int w, uint64_t p, int want_sai,
              pfp_parse_result *out);
void pfp_parse_result_free(pfp_parse_result *r);

/* ------------------------------------------------------------------------------------
 * Suffix sorting, drop-in for gsa/gsacak.h:78-105 (32-bit build, uint_t = uint32_t).
 *   pfp_sacak_int  == sacak_int(s,SA,n,k)   s[n-1]==0 unique smallest   (bwtparse.c:167)
 *   pfp_sacak      == sacak(s,SA,n)                                      (simplebwt.c:77)
 *   pfp_gsacak     == gsacak(s,SA,LCP,DA,n) with LCP==DA==NULL: separators (byte 1) ordered
 *                     by position, s[n-1]==0                             (pfbwt.cpp:495)
 * Return 0 on success (the reference returns the recursion depth, callers only test >=0).
 * ------------------------------------------------------------------------------------ */
int pfp_sacak_int(pfp_ctx *ctx, const uint32_t *s, uint32_t *SA, uint64_t n, uint64_t k);
int pfp_sacak(pfp_ctx *ctx, const uint8_t *s, uint32_t *SA, uint64_t n);
int pfp_gsacak(pfp_ctx *ctx, const uint8_t *s, uint32_t *SA, uint64_t n);
/* The same for the reference's -DM64 build (gsa/gsacak.h:42-60: uint_t = uint64_t, int_text stays 32 bits), which
 * bigbwt selects for parses / dictionaries / texts beyond the 32-bit limits (bigbwt:109-151, 177-194): 64-bit SA
 * entries, n up to 2^40.  Inside the library the index width follows the input size in every entry point
 * (32-bit positions below 4 GiB, 64-bit above; PFP_FORCE_IDX64=1 in the environment forces the wide build). */
/* gsacak with its optional outputs (gsa/gsacak.h:96-105; either may be NULL): LCP[i] = common prefix of the
 * suffixes SA[i-1], SA[i] with separators and the final 0 ending the count, LCP[0] = 0; DA[i] = index of the
 * string suffix SA[i] starts in (gsa/README.md:76-104).  The reference's pfbwt passes DA = NULL and uses LCP only
 * for its "same suffix as the entry before" test (pfbwt.cpp:204-209), which pfp_merge answers from rank equality. */
int pfp_gsacak_lcp_da(pfp_ctx *ctx, const uint8_t *s, uint32_t *SA, int32_t *LCP, int32_t *DA, uint64_t n);
int pfp_gsacak_lcp_da64(pfp_ctx *ctx, const uint8_t *s, uint64_t *SA, int64_t *LCP, int64_t *DA, uint64_t n);
int pfp_sacak_int64(pfp_ctx *ctx, const uint32_t *s, uint64_t *SA, uint64_t n, uint64_t k);
int pfp_sacak64(pfp_ctx *ctx, const uint8_t *s, uint64_t *SA, uint64_t n);
int pfp_gsacak64(pfp_ctx *ctx, const uint8_t *s, uint64_t *SA, uint64_t n);

/* ------------------------------------------------------------------------------------
 * Stage 2: bwtparse main (bwtparse.c:212-322): SA of the parse, BWT(P), inverted lists and
 * the permuted last / sai arrays.  Caller-allocated outputs: ilist u32[P+1],
 * bwlast u8[P+1], bwsai 5*(P+1) bytes (only when sai != NULL).
 * ------------------------------------------------------------------------------------ */
int pfp_bwtparse(pfp_ctx *ctx, const uint32_t *parse, uint64_t P, const uint8_t *last,
                 const uint8_t *sai /*5P bytes or NULL*/, const uint32_t *occ, uint64_t n_words,
                 uint32_t *ilist, uint8_t *bwlast, uint8_t *bwsai);

/* ------------------------------------------------------------------------------------
 * Stage 3: pfbwt main (pfbwt.cpp:320-418 -> bwt() :109-242, pfthreads.hpp:403-518).
 * n_plus_1 = ilist length = P+1.  Outputs library-allocated (pfp_free):
 *   bwt (.bwt) n+1 bytes; sa (.sa) 5n bytes; ssa/esa (.ssa/.esa) 10 bytes per pair.
 * ------------------------------------------------------------------------------------ */
typedef struct {
  uint8_t *bwt; uint64_t bwt_size;
  uint8_t *sa;  uint64_t sa_bytes;
  uint8_t *ssa; uint64_t ssa_bytes;
  uint8_t *esa; uint64_t esa_bytes;
} pfp_bwt_result;
int pfp_merge(pfp_ctx *ctx, const uint8_t *dict, uint64_t dict_size, const uint32_t *occ,
              uint64_t n_words, const uint32_t *ilist, const uint8_t *bwlast,
              const uint8_t *bwsai /*5(P+1) bytes or NULL*/, uint64_t n_plus_1, int w, int flags,
              pfp_bwt_result *out);
void pfp_bwt_result_free(pfp_bwt_result *r);

/* ------------------------------------------------------------------------------------
 * The whole chain in one call, as `bigbwt -w W -p M [-S|-s|-e] file` runs it (bigbwt:69-156),
 * with every intermediate kept in HBM (no files).  Host text in, host results out.
 * ------------------------------------------------------------------------------------ */
int pfp_bigbwt(pfp_ctx *ctx, const uint8_t *text, uint64_t n, int w, uint64_t p, int flags,
               pfp_bwt_result *out);

/* File to files: like pfp_bigbwt, but the outputs are streamed from HBM into <base>.bwt and, as the flags ask,
 * <base>.sa / .ssa / .esa (created or truncated) instead of being returned: what the `bigbwt` driver calls.  `text`
 * may be an mmap of the input file: it is read once, front to back, in chunks.  out_bytes (may be NULL) = the sizes
 * written {bwt, sa, ssa, esa}.
 * An output of 64 MB or more whose file lies in a memory file system (tmpfs: /dev/shm) is not staged through pinned buffers and
 * pwrite(): the file is created at its final size, mapped, its pages allocated and registered with the runtime by a helper
 * thread beside the text input and the chain, and the result copied from HBM straight into them (.bwt and .sa, whose sizes n
 * fixes, from the start of the call; .ssa / .esa once their run count is known).  Consequences a caller can see: those files
 * exist (zero-filled) while the call runs, and are removed again if it fails; PFP_MAP_OUTPUT=0 in the environment keeps every
 * output on the pwrite path (the reference writes with fwrite / pwrite: pfbwt.cpp:145-223, pfthreads.hpp:369-376). */
int pfp_bigbwt_files(pfp_ctx *ctx, const uint8_t *text, uint64_t n, int w, uint64_t p, int flags,
                     const char *base, uint64_t out_bytes[4]);
/* The same with the text taken from bytes [file_offset, file_offset + n) of an open file descriptor instead of a host buffer:
 * a few threads pread() straight into the pinned staging buffers (an mmap'ed input costs a page fault per 4 KB: 2 GB/s for a
 * 12.6 GB file in /dev/shm, where this reads at the rate of the PCIe link).  What host/bigbwt.c calls for a plain input file
 * (the reference's parsers read theirs with fread / getc: newscan.cpp:355-377). */
int pfp_bigbwt_fd(pfp_ctx *ctx, int fd, uint64_t file_offset, uint64_t n, int w, uint64_t p, int flags, const char *out_base,
                  uint64_t out_bytes[4]);

/* Device-resident variant: d_text is a device pointer to n bytes; d_bwt must hold n+1 bytes.
 * Optional device outputs (may be NULL unless the flag is set):
 *   d_sa   u64[n+1]  SA value per BWT position (d_sa[0] = n), flags & (SA|SSA|ESA).  With PFP_FLAG_SA every entry is
 *                    written.  With only PFP_FLAG_SSA / PFP_FLAG_ESA the entries at the run boundaries of the BWT -
 *                    positions j with BWT[j] != BWT[j-1] or BWT[j] != BWT[j+1], j = 0 and j = n, i.e. every entry
 *                    the .ssa/.esa files hold (pfbwt.cpp:605-676) - are written and the rest is left untouched.
 * Run-sampled / packed outputs are derived from d_bwt/d_sa by pfp_pack5_dev / pfp_sample_runs_dev below.
 * *n_used returns the parsed length; bwt length is *n_used + 1. */
int pfp_bigbwt_dev(pfp_ctx *ctx, const void *d_text, uint64_t n, int w, uint64_t p, int flags,
                   void *d_bwt, void *d_sa, uint64_t *n_used);

/* Reference file formats from device-resident results (device pointers in and out):
 *   pfp_pack5_dev       : count u64 values -> 5-byte little-endian ints (utils.c:112-129; `.sa`, pfbwt.cpp:159-160:
 *                         pass d_sa + 1 and count = n, SA[0] = n is not written)
 *   pfp_sample_runs_dev : the `.ssa` (run_end = 0: positions j with BWT[j] != BWT[j-1], incl. j = 0; pfbwt.cpp:169-174,
 *                         184-189, 605-676) or `.esa` (run_end = 1: BWT[j] != BWT[j+1], incl. j = n; pfbwt.cpp:175-179,
 *                         225-229) pairs <j, SA[j]>, 5 + 5 bytes each, of the BWT slice [pos_base, pos_base + count):
 *                         d_bwt / d_sa point at the slice's first element; left_byte / right_byte = the BWT byte just
 *                         before / after the slice (a rank's halo from its neighbours), -1 at the ends of
 *                         the whole BWT.  *n_pairs = boundaries in the slice; d_out10 == NULL only counts; more pairs
 *                         than cap_pairs -> PFP_ELIMIT (with *n_pairs set).  Concatenating the slices' outputs in
 *                         order gives the reference's file. */
int pfp_pack5_dev(pfp_ctx *ctx, const void *d_vals_u64, uint64_t count, void *d_out5);
/* writes nbytes of device memory into `path` at file_offset (file created if missing, never truncated), streamed
 * through pinned staging buffers: how a rank of the multi-GPU chain stores its slice of .bwt/.sa/.ssa/.esa -
 * the reference's threads pwrite() their ranges the same way (pfthreads.hpp:369-376) */
int pfp_pwrite_dev(pfp_ctx *ctx, const char *path, uint64_t file_offset, const void *d_src, uint64_t nbytes);
int pfp_sample_runs_dev(pfp_ctx *ctx, const void *d_bwt, const void *d_sa, uint64_t count, uint64_t pos_base,
                        int left_byte, int right_byte, int run_end, void *d_out10, uint64_t cap_pairs,
                        uint64_t *n_pairs);

/* Device-resident chain that hands back the reference's SA-derived FILES instead of SA values: d_out[0] = .sa
 * bytes (PFP_FLAG_SA), d_out[1] = .ssa, d_out[2] = .esa - device buffers allocated by the library, released with
 * pfp_dev_free; out_bytes their sizes; entries for flags not set stay NULL / 0.  The SA values live inside the call
 * only (-S: 8 bytes per text byte, allocated after the suffix sorter has returned its scratch; -s / -e: 8 bytes per
 * run boundary of the BWT), so a >= 10 GB input with -s fits one GPU.  d_bwt as in pfp_bigbwt_dev.  The buffers are
 * blocks of the context's memory pool: pfp_dev_free hands them back for reuse by later calls on the context, so the
 * caller must have finished reading them (or have read them on the context's stream) before it frees them; they are
 * released with the context at the latest. */
int pfp_bigbwt_formats_dev(pfp_ctx *ctx, const void *d_text, uint64_t n, int w, uint64_t p, int flags, void *d_bwt,
                           void *d_out[3], uint64_t out_bytes[3], uint64_t *n_used);
void pfp_dev_free(pfp_ctx *ctx, void *d_ptr);
/* device -> host copy through the context's pinned staging buffers (for buffers the library handed out) */
int pfp_memcpy_d2h(pfp_ctx *ctx, void *host_dst, const void *d_src, uint64_t nbytes);

/* per-call statistics of the most recent pfp_bigbwt / pfp_bigbwt_dev / pfp_parse on this ctx */
typedef struct {
  uint64_t n, n_phrases, n_words, dict_size;
  uint64_t sa_rounds_dict, sa_rounds_parse;
  uint64_t hard_groups, hard_chars;
  uint64_t hard_big_groups, hard_max_chars, hard_max_members;
  uint64_t hash_reseeds;
  uint64_t extra_triggers;   /* window hashes added by the fused chain to split giant phrases */
  uint64_t index_bits;       /* 32 or 64: width of dictionary positions / suffix-array slots used (bigbwt:130-151) */
  uint64_t hard_minor_groups, hard_minor_chars; /* hard groups done by majority fill; occurrences ranked for them */
  double ms_scan, ms_phrases, ms_sa_dict, ms_sa_parse, ms_merge, ms_total; /* host wall, synced */
  double parse_density;      /* fused chain: the text was cut with probability parse_density / p (pfp_set_parse_density) */
} pfp_stats;
int pfp_get_stats(const pfp_ctx *ctx, pfp_stats *st);
/* when set (default 0) every pipeline phase is bracketed by a stream sync so ms_* are filled */
void pfp_set_profiling(pfp_ctx *ctx, int on);
/* Per-kernel device times measured with HIP events on the ctx stream.  pfp_set_kernel_trace(ctx,1)
 * clears the table and starts recording; pfp_get_kernel_trace synchronises, resolves the events
 * and returns the number of rows (rows beyond cap are counted, not written).  algo_bytes is the
 * sum over the launches of the kernel's algorithmic bytes (DESIGN.md "kernels"). */
typedef struct { char name[64]; uint64_t launches; double total_ms; uint64_t algo_bytes; } pfp_kernel_stat;
void pfp_set_kernel_trace(pfp_ctx *ctx, int on);
int pfp_get_kernel_trace(pfp_ctx *ctx, pfp_kernel_stat *out, int cap);

/* Fused chain only (pfp_bigbwt / pfp_bigbwt_dev): phrases longer than max_phrase bytes are split
 * by adding a few extra trigger windows taken from inside them (default 32768; 0 = parse exactly
 * as the reference does).  The .bwt/.sa/.ssa/.esa outputs do not depend on the parse
 * (SURVEY.md 2.2-Q11); pfp_scan / pfp_parse always use the reference's trigger set. */
/* Diagnostic (tests): the hand-written first-round sort (csrc/radix.hip: what replaces the bucket passes of gsacak.c:1395-1524
 * in the first round of the suffix sorter) on caller data: keys, and 32-bit values if vals != NULL, sorted in place, stable on
 * key bits [lo, hi). */
int pfp_debug_msd_sort(pfp_ctx *ctx, uint64_t *keys, uint32_t *vals, uint64_t n, int lo, int hi);
void pfp_set_max_phrase(pfp_ctx *ctx, uint64_t max_phrase);
/* Fused chain only: which function of the last w bytes cuts the text.  fast != 0 (default): a multiply-add hash of the window,
 * a third of the arithmetic of the reference's `KR_window` (newscan.cpp:168-202: mod 1999999973, then mod p) with the same 1 / p
 * density; 0 (or PFP_WINDOW_HASH=kr in the environment at pfp_ctx_create): Karp-Rabin as in the reference.  The outputs do not
 * depend on the choice (quirk Q1 is reproduced either way); pfp_scan / pfp_parse / the stage executables
 * always cut exactly where the reference does.  With fast == 0 and max_phrase == 0 the fused chain parses like the reference. */
void pfp_set_window_hash(pfp_ctx *ctx, int fast);
/* Fused chain with the window hash only: cut with probability density / p instead of 1 / p.  The outputs do not depend on it; the
 * work does - for c copies at mutation rate r the dictionary grows with the phrase length (about G (1 + c r L) bytes) while the
 * parse shrinks (n / L phrases), so a collection of many near-identical copies is processed faster, and in half the memory, with
 * shorter phrases (density 2 = what -p p/2 would parse like), and a single genome is not.  density = 0 (default; PFP_PARSE_DENSITY
 * in the environment at pfp_ctx_create sets another): the chain decides between 1 and p / 48 (phrases of ~48 bytes) itself - one
 * scan at the higher density, a content-defined sample of the cuts, all cuts kept if the sampled 64-byte contexts show more
 * variants than loci (weighted by the phrase length), else the cuts beyond 1 / p dropped again (scan.hip: choose_parse_density).  density = 1 pins what -p says; the staged entry points and the stage
 * executables always parse exactly like the reference.  pfp_stats.parse_density tells what a call used; `bigbwt --density D`. */
int pfp_set_parse_density(pfp_ctx *ctx, double density);
/* Index width of dictionary positions and suffix-array slots: 0 = by size (32 bits below 4 GiB of dictionary /
 * text, 64 above: the reference's choice between its 32-bit and -DM64 executables, bigbwt:109-151), 64 = always
 * the wide build (what PFP_FORCE_IDX64=1 in the environment sets at pfp_ctx_create), 32 = the narrow build wherever its positions
 * fit.  By size means: narrow below 2^31 bytes of dictionary, wide from 2^32 - 16 on, and in between the wide build where ~96 bytes
 * of device memory per dictionary byte are free (it keeps the sorter's pivot rounds there; the narrow build has no spare bit in a
 * position for them), else the narrow one.  Outputs are identical. */
int pfp_set_index_bits(pfp_ctx *ctx, int bits);

/* ------------------------------------------------------------------------------------
 * Multi-GPU chain, one rank's share (the reference's analogue is the byte-range
 * threading of pscan.hpp:114-165 and the output-range threading of pfthreads.hpp:456-493).
 * The caller (big-bwt_amd/dist.py, torch.distributed over RCCL) shards the text, moves the
 * halos and runs the allgathers between the steps; all pointers are device pointers.
 *   pfp_dist_propose_triggers: window hashes (<= 8) that would split this shard's giant phrases
 *       (pfp_set_max_phrase); the caller allgathers them, every rank passes the union as
 *       extra_hashes so that all ranks parse with one trigger set (outputs do not depend on it)
 *   pfp_dist_local_parse : d_text = halo (the last halo_len bytes of the previous shard; 0 for the
 *       first rank) followed by this rank's shard, n bytes in all; global_offset = position of the
 *       shard's first byte in the whole text.  Owns the phrases that end inside the shard.  want_sai = the output
 *       flags of the run (PFP_FLAG_*; 0 = BWT only: no sa info is kept).
 *       out_sizes = {local dict bytes, local words, local phrases, local position of last trigger}
 *   pfp_dist_export_local: copies the local dictionary (words + 0x01), its occ (u32), last (u8)
 *       and sai (u64) into caller buffers (any may be NULL)
 *   pfp_dist_global      : d_union = the ranks' local dictionaries back to back, d_union_occ their
 *       occ; my_word_base = index of this rank's first word in the union.  Builds the global
 *       dictionary and writes this rank's parse as global 1-based ranks (u32[local phrases]).
 *       out_info = {global words, global dict bytes, doubling rounds}
 *   pfp_dist_global_sort / pfp_dist_global_finish: the same in two steps, with the suffix array of
 *       the global dictionary sharded by key range (the reference shards the same array by index
 *       range across threads, pfthreads.hpp:171-176): share `part` of `parts` sorts the suffixes
 *       whose first-round key falls in its part of the key space (splitters from a deterministic
 *       sample, identical on all ranks, no exchange) and holds one contiguous range of SA(D).
 *       d_wslot_out (u64[n_union], the first out_info[0] used): 1 + SA(D) slot of every global
 *       word's first suffix if this share holds it, else 0.  out_info = {global words, global
 *       dict bytes, sorting rounds, complete (0/1), slots held, first slot, BWT positions the held
 *       slots emit, index width used (32 / 64)}.  complete == 0: some group could not be settled without other shares'
 *       ranks - every rank must then redo the step with parts = 1 (the replicated sort).
 *       The caller allgathers d_wslot_out (and complete / emit counts) and passes all `parts`
 *       arrays, out_info[0] entries each, back to back to pfp_dist_global_finish, which ranks the
 *       words and writes this rank's parse.  parts = 1 needs no exchange (= pfp_dist_global).
 *   pfp_dist_merge       : d_sym/d_last/d_sai = the whole parse in text order (all ranks);
 *       n_total = text length; emits BWT positions [out_lo,out_hi) into d_bwt_slice (u8) and, with
 *       flags, SA values into d_sa_slice (u64).  After a sharded sort [out_lo,out_hi) must be the
 *       range the held slots emit: out_lo = sum of the emit counts of the lower shares.
 * ------------------------------------------------------------------------------------ */
int pfp_dist_propose_triggers(pfp_ctx *ctx, const void *d_text, uint64_t n, int w, uint64_t p,
                              uint32_t out_hashes[8], uint32_t *n_hashes);
int pfp_dist_local_parse(pfp_ctx *ctx, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p,
                         int is_first, int is_last, uint64_t global_offset, int want_sai,
                         const uint32_t *extra_hashes, uint32_t n_extra, uint64_t out_sizes[4]);
/* Round 4: the multi-GPU chain under a PARSE PLAN - the window hash and the phrase length by repetitiveness of the fused chain
 * (pfp_set_window_hash, pfp_set_parse_density), agreed between the ranks.  plan[4]: [0] 0 = the reference's Karp-Rabin hash (what
 * the two calls above cut by), 1 = the window hash; [1] its seed; [2] the density (the bits of a double: cuts with probability
 * density / p); [3] 1 while that density is a candidate the ranks still have to decide on.
 *   pfp_dist_parse_plan        rank 0, from the text's first bytes (host): the plan every rank gets, and the first window's hash
 *                              under it (banned as an extra trigger: SURVEY.md 2.2-Q1).  The density is a candidate (p / 48) on one
 *                              or two ranks and nominal beyond: every rank reads the whole parse, only the dictionary is shared
 *   pfp_dist_propose_triggers2 with plan[3] set: this rank's sample of its cuts (sorted 64-bit context hashes of the cuts at or
 *                              after halo_len, at most sample_cap, in d_sample) and no proposals
 *   -- the hosts all-gather the samples --
 *   pfp_dist_decide_density    on every rank, the same gathered samples: settles plan[2] (the candidate, or 1) and clears plan[3]
 *   pfp_dist_propose_triggers2 under the settled plan: as pfp_dist_propose_triggers (the proposals are made at the density the parse
 *                              will have: a window of a periodic stretch can cut at the candidate density and not at the nominal one)
 *   pfp_dist_local_parse2      as pfp_dist_local_parse, under the settled plan (extra_hashes are hashes under the plan) */
int pfp_dist_parse_plan(pfp_ctx *ctx, const uint8_t *first_bytes, uint64_t n_bytes, int w, uint64_t p, uint32_t ranks, uint64_t plan[4],
                        uint64_t *first_hash);
int pfp_dist_propose_triggers2(pfp_ctx *ctx, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p, const uint64_t plan[4],
                               uint32_t out_hashes[8], uint32_t *n_hashes, void *d_sample, uint64_t sample_cap, uint64_t *n_sample);
int pfp_dist_decide_density(pfp_ctx *ctx, const void *d_samples, uint64_t count, uint64_t p, uint64_t plan[4]);
int pfp_dist_local_parse2(pfp_ctx *ctx, const void *d_text, uint64_t n, uint64_t halo_len, int w, uint64_t p,
                          int is_first, int is_last, uint64_t global_offset, int want_sai, const uint64_t plan[4],
                          const uint32_t *extra_hashes, uint32_t n_extra, uint64_t out_sizes[4]);
int pfp_dist_export_local(pfp_ctx *ctx, void *d_dict, void *d_occ, void *d_last, void *d_sai);
int pfp_dist_global(pfp_ctx *ctx, const void *d_union, uint64_t union_bytes, const void *d_union_occ,
                    uint64_t n_union, uint64_t my_word_base, void *d_sym_out, uint64_t out_info[3]);
int pfp_dist_global_sort(pfp_ctx *ctx, const void *d_union, uint64_t union_bytes, const void *d_union_occ,
                         uint64_t n_union, uint32_t part, uint32_t parts, void *d_wslot_out, uint64_t out_info[8]);
int pfp_dist_global_finish(pfp_ctx *ctx, const void *d_wslot_all, uint32_t parts, uint64_t my_word_base,
                           void *d_sym_out);
/* Hash-partitioned deduplication (the exchange SURVEY.md 8e calls A; the reference's threaded parser shards its
 * maps by `hash % (3 N)`, pscan.cpp:137-205): every distinct word is owned by the rank its identity hash points
 * at, so the union of the local dictionaries is deduplicated in `parts` disjoint pieces and only distinct words
 * are gathered.  Between the calls the caller runs an all-to-all of (words, occ), an all-to-all of the answers
 * and an allgatherv of the owners' distinct words (dist.py):
 *   pfp_dist_partition_words      : counts[2*o], counts[2*o+1] = words / bytes (one 0x01 per word included) this
 *                                   rank sends to owner o
 *   pfp_dist_export_partition     : the local words (each + 0x01) and their occ, grouped by owner, owner 0 first
 *   pfp_dist_owner_dedup          : d_bytes/d_occ = what all ranks sent to this owner, back to back in rank order;
 *                                   d_pid_out[u] = index of received word u among this owner's distinct words;
 *                                   out = {distinct words, their bytes}
 *   pfp_dist_export_owned         : the owner's distinct words (each + 0x01) and their summed occ
 *   pfp_dist_global_sort_distinct : as pfp_dist_global_sort, on the gathered owner pieces (owner 0 first, no
 *                                   further dedup); d_gid_sent[k] = global id (owner base + pid answer) of the k-th
 *                                   word this rank exported; pfp_dist_global_finish then needs no my_word_base */
int pfp_dist_partition_words(pfp_ctx *ctx, uint32_t parts, uint64_t *counts);
int pfp_dist_export_partition(pfp_ctx *ctx, void *d_bytes, void *d_occ);
int pfp_dist_owner_dedup(pfp_ctx *ctx, const void *d_bytes, uint64_t nbytes, const void *d_occ, uint64_t n_words,
                         void *d_pid_out, uint64_t out[2]);
int pfp_dist_export_owned(pfp_ctx *ctx, void *d_bytes, void *d_occ);
int pfp_dist_global_sort_distinct(pfp_ctx *ctx, const void *d_dict, uint64_t dict_bytes, const void *d_occ, uint64_t n_words,
                                  const void *d_gid_sent, uint32_t part, uint32_t parts, void *d_wslot_out,
                                  uint64_t out_info[8]);
/* The suffix array of the (replicated) parse in shares, like the dictionary's: pfp_dist_parse_sort sorts the suffixes of the whole
 * parse d_sym (u32[P], all ranks') whose first-round key lies in share `part` of `parts` into d_sa_out (u32, room for P + 1) -
 * out_info = {entries, first slot, complete, rounds}; complete = 0: only a doubling round could go on (then, or if any rank says so,
 * nobody sets anything and pfp_dist_merge sorts the whole parse itself as bwtparse.c does).  The caller all-gathers the shares in
 * rank order (they are consecutive ranges of the array) and hands the P + 1 entries to pfp_dist_set_parse_sa before
 * pfp_dist_merge, which uses them once. */
int pfp_dist_parse_sort(pfp_ctx *ctx, const void *d_sym, uint64_t P, uint32_t part, uint32_t parts, void *d_sa_out, uint64_t out_info[4]);
int pfp_dist_set_parse_sa(pfp_ctx *ctx, const void *d_sa, uint64_t count);
int pfp_dist_merge(pfp_ctx *ctx, const void *d_sym, uint64_t P, const void *d_last, const void *d_sai, int flags,
                   uint64_t n_total, uint64_t out_lo, uint64_t out_hi, void *d_bwt_slice, void *d_sa_slice);
/* After pfp_dist_merge with PFP_FLAG_SSA / PFP_FLAG_ESA and d_sa_slice == NULL (the SA values then stay inside: 8 bytes
 * per run boundary of the slice instead of 8 per position): the slice's pieces of .ssa (run_end == 0) / .esa
 * (run_end != 0) as 10-byte pairs <global position, SA value> (pfbwt.cpp:605-676), written from the run maps the merge
 * left - no pass over the BWT bytes.  drop_edge: the slice's first (.ssa) / last (.esa) position is not a run start /
 * end after all, because the neighbouring slice's adjacent byte is the same (the caller has exchanged those bytes).
 * d_out10 == NULL: count only. */
int pfp_dist_sample_runs(pfp_ctx *ctx, int run_end, int drop_edge, void *d_out10, uint64_t cap_pairs, uint64_t *n_pairs);
void pfp_dist_release(pfp_ctx *ctx);

/* ------------------------------------------------------------------------------------
 * One BWT on n_dev GPUs of one node, from one process (csrc/multi.hip): a host thread and a context per device run
 * the chain above and meet in RCCL collectives over xGMI (grouped ncclSend / ncclRecv of the ranks' pieces; librccl is
 * loaded on the first call).  The reference's analogue is its threaded build, `bigbwt -t N`: pscan.cpp / pscan.hpp:114-165
 * (byte ranges of the input, hash-sharded dictionary) and pfthreads.hpp:171-176, 369-376, 456-493 (the suffix array sharded
 * by range, output ranges written with pwrite).  text = the whole input in host memory (rank r reads bytes
 * [n r / n_dev, n (r+1) / n_dev)); halo = bytes of a range its right neighbour also reads, must cover the longest phrase
 * (0 = 1 MiB); outputs out_base.bwt / .sa / .ssa / .esa as pfp_bigbwt_files writes them.  A failure on any rank ends all
 * ranks; its text goes to errbuf.  Bytes <= 2 in the text are an error here (PFP_EFORMAT), not the end of the input.
 * PFP_MULTI_LOOPBACK=1 (tests on a one-GPU box): the ranks share the devices given, modulo the visible ones, and exchange
 * through device copies instead of RCCL. */
typedef struct {
  uint64_t n, n_words, n_phrases, dict_size, index_bits;
  uint64_t ranks, sa_shares;      /* sa_shares = 1: a key range could not finish alone, every rank sorted the whole dictionary */
  uint64_t parse_shares;          /* ranks when the parse's suffix array was sorted in key ranges too, 1 when every rank sorted all of it */
  double ms_chain, ms_total;      /* rank 0: upload to finished device outputs; + files */
  double parse_density;           /* the plan's settled density (1 under the Karp-Rabin plan) */
} pfp_multi_stats;
int pfp_bigbwt_files_multi(int n_dev, const int *devices, const uint8_t *text, uint64_t n, int w, uint64_t p, int flags,
                           uint64_t halo, const char *out_base, pfp_multi_stats *stats, char *errbuf, uint64_t errbuf_len);

/* The RCCL transport of pfp_bigbwt_files_multi on ONE device (tests on a one-GPU box): librccl resolved with dlopen, a communicator
 * from ncclCommInitAll over `device`, every exchange shape of the chain as a self send / recv inside a group, and the one-collective
 * all-gather; returns PFP_OK when every byte came back. */
int pfp_multi_rccl_selftest(int device, char *errbuf, uint64_t errbuf_len);
/* the same, then (inject_failure != 0) an exchange that fails between ncclGroupStart and ncclGroupEnd with a send already queued:
 * the group is closed on the way out and every communicator aborted (ncclCommAbort), as a failing rank of the chain does; PFP_OK =
 * it failed as intended and came back */
int pfp_multi_rccl_selftest2(int device, int inject_failure, char *errbuf, uint64_t errbuf_len);

/* ---- micro entry points used by bench.py's roofline leg and by the parity tests ---- */
/* copy a device-resident text into the ctx's padded staging buffer (T' = Dollar.T.Dollar^w) */
int pfp_stage_text_dev(pfp_ctx *ctx, const void *d_text, uint64_t n, int w);
/* run only stage 1a (K1 window-hash + trigger mask, block-count scan, K2 compaction) on the
 * staged text; all work is enqueued on pfp_ctx_stream(ctx) and finished on return. */
int pfp_scan_staged(pfp_ctx *ctx, uint64_t p, uint64_t *n_ends);
/* enqueue only K1 (the window-hash kernel, n + n/8 bytes of traffic) - no sync, for event timing */
int pfp_scan_k1_enqueue(pfp_ctx *ctx, uint64_t p);

#ifdef __cplusplus
}
#endif
#endif
