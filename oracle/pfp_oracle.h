/* pfp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of alshai/Big-BWT's parse -> SA -> BWT path, used as the parity
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
 * the product (big-bwt_amd/, the C-ABI library, the bigbwt driver) may include, link or call
 * this.  Pinned against the real reference binaries (oracle/_ref, built by oracle/Makefile
 * from /root/reference) and against the committed fixtures under tests/golden/.
 *
 * Every function names the reference file:line it restates.
 */
#ifndef PFP_ORACLE_H
#define PFP_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_DOLLAR 2      /* utils.h:6 */
#define ORC_ENDOFWORD 1   /* utils.h:7 */
#define ORC_ENDOFDICT 0   /* utils.h:8 */

#define ORC_FLAG_SA   1   /* -S full suffix array             (pfbwt.cpp:159-160) */
#define ORC_FLAG_SSA  2   /* -s run-start sampled SA          (pfbwt.cpp:169-174) */
#define ORC_FLAG_ESA  4   /* -e run-end sampled SA            (pfbwt.cpp:175-179) */

/* stage 1 outputs (newscan.cpp:569-650): byte formats of SURVEY.md 2.3, sai unpacked to u64 */
typedef struct {
  uint64_t n_used;      /* bytes of text actually parsed (stops at first byte <= 2, newscan.cpp:364) */
  uint8_t *dict;        /* .dict: sorted phrases, each + 0x01, final 0x00 */
  uint64_t dsize;
  uint32_t *occ;        /* .occ */
  uint32_t d;           /* # distinct phrases */
  uint32_t *parse;      /* .parse: 1-based lexicographic ranks, text order */
  uint64_t P;           /* # phrases */
  uint8_t *last;        /* .last */
  uint64_t *sai;        /* .sai (values; the file packs each into 5 LE bytes) */
  uint64_t *phash;      /* .parse_old: 64-bit KR hash per phrase (newscan.cpp:290) */
} orc_parse_t;

typedef struct {
  uint8_t *bwt;  uint64_t nbwt;        /* .bwt, n+1 bytes */
  uint64_t *sa;  uint64_t nsa;         /* .sa values (n entries, SA[0]=n omitted: pfbwt.cpp:158-162) */
  uint64_t *ssa; uint64_t nssa;        /* .ssa as (pos,sa) pairs, nssa = # pairs */
  uint64_t *esa; uint64_t nesa;        /* .esa as (pos,sa) pairs */
  uint64_t full_words, easy, hard;     /* pfbwt.cpp:231-233 counters */
} orc_bwt_t;

/* newscan.cpp:168-202 KR_window: hash of exactly w bytes */
uint64_t orc_kr_window(const uint8_t *win, int w);
/* newscan.cpp:229-239 kr_hash */
uint64_t orc_kr_hash(const uint8_t *s, uint64_t len);
/* newscan.cpp:363-377 trigger scan only: returns # of phrase ends written (malloc'ed *ends, T positions) */
int orc_scan(const uint8_t *text, uint64_t n, int w, uint64_t p, uint64_t **ends, uint64_t *nends);
/* newscan.cpp main: process_file + sort + writeDictOcc + remapParse */
int orc_parse(const uint8_t *text, uint64_t n, int w, uint64_t p, orc_parse_t *out);
void orc_parse_free(orc_parse_t *o);

/* gsa/gsacak.h:78-105 semantics (32-bit build): SA of s[0..n-1], s[n-1]==0 unique smallest */
int orc_sacak(const uint8_t *s, uint32_t *SA, uint64_t n);
int orc_sacak_int(const uint32_t *s, uint32_t *SA, uint64_t n, uint64_t k);
/* separators (byte 1) ordered by position; LCP (may be NULL) stops at separators */
int orc_gsacak(const uint8_t *s, uint32_t *SA, int32_t *LCP, uint64_t n);

/* bwtparse.c:212-322; sai may be NULL; outputs caller-allocated: ilist[P+1], bwlast[P+1], bwsai[P+1] */
int orc_bwtparse(const uint32_t *parse, uint64_t P, const uint8_t *last, const uint64_t *sai,
                 const uint32_t *occ, uint32_t d, uint32_t *ilist, uint8_t *bwlast, uint64_t *bwsai);

/* pfbwt.cpp:109-242 (single thread path) */
int orc_pfbwt(const uint8_t *dict, uint64_t dsize, const uint32_t *occ, uint32_t d,
              const uint32_t *ilist, const uint8_t *bwlast, const uint64_t *bwsai, uint64_t P,
              int w, int flags, orc_bwt_t *out);
void orc_bwt_free(orc_bwt_t *o);

/* bigbwt:69-156 chain newscanNT -> bwtparse -> pfbwtNT */
int orc_bigbwt(const uint8_t *text, uint64_t n, int w, uint64_t p, int flags, orc_bwt_t *out);
/* simplebwt.c:28-100: bwt must hold n+1 bytes */
int orc_simplebwt(const uint8_t *text, uint64_t n, uint8_t *bwt);

/* utils.c:112-129 5-byte little-endian ints */
void orc_pack5(const uint64_t *v, uint64_t cnt, uint8_t *out);
void orc_unpack5(const uint8_t *in, uint64_t cnt, uint64_t *v);

/* SURVEY.md section 4 GEN: synthetic repetitive FASTA; returns bytes written (0 on overflow) */
uint64_t orc_gen_fasta(uint64_t G, uint32_t C, double r, uint64_t seed, const uint64_t *nblk, uint32_t nnblk,
                       uint8_t *out, uint64_t cap);

/* newscan.cpp:332-352 (-f) over kseq.h:178-222: the text the parser sees for a FASTA/FASTQ file.
 * out must hold n bytes; returns the bytes written. */
uint64_t orc_fasta_text(const uint8_t *in, uint64_t n, uint8_t *out);

void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
