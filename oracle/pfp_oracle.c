/* pfp_oracle.c -- TEST INFRASTRUCTURE ONLY (see pfp_oracle.h).
 *
 * CPU restatement, in plain C, of the parse -> SA -> BWT path of alshai/Big-BWT.
 * Written from the behaviour described in SURVEY.md and read from the reference sources;
 * no reference text is reproduced.  Suffix sorting is a simple prefix-doubling sorter of our
 * own (the reference uses gSACA-K, gsa/gsacak.c); by SURVEY.md 2.2-Q11 the suffix array of
 * a string with a unique smallest terminator is unique, so any correct sorter gives the same
 * arrays the reference gets (validated against oracle/_ref in tests/test_oracle_vs_ref.py).
 *
 * Parity status: PINNED -- against the reference binaries built from /root/reference
 * (oracle/_ref) and the fixtures they generated (tests/golden/, tests/golden/make_golden.py).
 */
#define _GNU_SOURCE
#include "pfp_oracle.h"
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <assert.h>

#define KR_PRIME   1999999973ULL          /* newscan.cpp:172 */
#define WORD_PRIME 27162335252586509ULL   /* newscan.cpp:232 */

void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------ hashes */

/* newscan.cpp:168-202: after >= w calls of addchar() the state equals
 * sum c[j]*256^(w-1-j) mod prime; we evaluate that closed form. */
uint64_t orc_kr_window(const uint8_t *win, int w) {
  uint64_t h = 0;
  for (int i = 0; i < w; i++) h = (h * 256 + win[i]) % KR_PRIME;
  return h;
}

/* newscan.cpp:229-239 */
uint64_t orc_kr_hash(const uint8_t *s, uint64_t len) {
  uint64_t h = 0;
  for (uint64_t k = 0; k < len; k++) h = (256 * h + s[k]) % WORD_PRIME;
  return h;
}

/* utils.c:112-129 */
void orc_pack5(const uint64_t *v, uint64_t cnt, uint8_t *out) {
  for (uint64_t i = 0; i < cnt; i++)
    for (int b = 0; b < 5; b++) out[5 * i + b] = (uint8_t)(v[i] >> (8 * b));
}
void orc_unpack5(const uint8_t *in, uint64_t cnt, uint64_t *v) {
  for (uint64_t i = 0; i < cnt; i++) {
    uint64_t x = 0;
    for (int b = 0; b < 5; b++) x |= (uint64_t)in[5 * i + b] << (8 * b);
    v[i] = x;
  }
}

/* length of the prefix the reference actually reads: it stops at the first byte <= Dollar
 * (newscan.cpp:364) */
static uint64_t usable_len(const uint8_t *t, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) if (t[i] <= ORC_DOLLAR) return i;
  return n;
}

/* ------------------------------------------------------------------ stage 1a: trigger scan */

/* newscan.cpp:363-377 with KR_window::addchar (newscan.cpp:194-202) kept as the rolling
 * update (window zero-initialised), so the first w-1 partial-window hashes are exactly the
 * reference's; a trigger is honoured only when the current word is longer than w
 * (save_update_word, newscan.cpp:248). */
int orc_scan(const uint8_t *text, uint64_t n, int w, uint64_t p, uint64_t **ends_out, uint64_t *nends) {
  if (w < 1 || p < 1) return -1;
  n = usable_len(text, n);
  uint64_t cap = n / (p > 4 ? p / 4 : 1) + 16, cnt = 0;
  uint64_t *ends = malloc(cap * sizeof *ends);
  int *window = calloc((size_t)w, sizeof *window);
  uint64_t asize_pot = 1;
  for (int i = 1; i < w; i++) asize_pot = (asize_pot * 256) % KR_PRIME;
  uint64_t hash = 0, tot = 0, wordlen = 1; /* word starts as "Dollar" */
  for (uint64_t i = 0; i < n; i++) {
    int c = text[i];
    int k = (int)(tot++ % (uint64_t)w);
    hash += KR_PRIME - ((uint64_t)window[k] * asize_pot) % KR_PRIME;
    hash = (256 * hash + (uint64_t)c) % KR_PRIME;
    window[k] = c;
    wordlen++;
    if (hash % p == 0 && wordlen > (uint64_t)w) {
      if (cnt == cap) { cap *= 2; ends = realloc(ends, cap * sizeof *ends); }
      ends[cnt++] = i;
      wordlen = (uint64_t)w;
    }
  }
  free(window);
  *ends_out = ends; *nends = cnt;
  return 0;
}

/* ------------------------------------------------------------------ stage 1: parse */

typedef struct { uint64_t hash; uint64_t off; uint32_t len; uint32_t occ; uint32_t rank; } wstat;

static const uint8_t *g_tp;   /* T' = Dollar . T . Dollar^w, used by the comparators */

static int cmp_words(const void *a, const void *b) {
  const wstat *x = *(const wstat *const *)a, *y = *(const wstat *const *)b;
  uint32_t m = x->len < y->len ? x->len : y->len;
  int c = memcmp(g_tp + x->off, g_tp + y->off, m);   /* unsigned bytes == std::string order */
  if (c) return c;
  return (x->len > y->len) - (x->len < y->len);
}

void orc_parse_free(orc_parse_t *o) {
  free(o->dict); free(o->occ); free(o->parse); free(o->last); free(o->sai); free(o->phash);
  memset(o, 0, sizeof *o);
}

/* newscan.cpp:569-650.  Phrase k = T'[s_k .. e_k]; consecutive phrases overlap by w bytes.
 * dedup by the 64-bit hash with a byte check (newscan.cpp:271-287, collision => error),
 * lexicographic sort (newscan.cpp:622-636), dict/occ (394-441), rank remap (443-466). */
int orc_parse(const uint8_t *text, uint64_t n, int w, uint64_t p, orc_parse_t *out) {
  memset(out, 0, sizeof *out);
  if (w < 4 || p < 10) return -1;                       /* newscan.cpp:537-544 */
  uint64_t *ends = NULL, ne = 0;
  n = usable_len(text, n);
  if (orc_scan(text, n, w, p, &ends, &ne)) return -1;
  uint64_t P = ne + 1;
  uint8_t *tp = malloc(n + (uint64_t)w + 1);
  tp[0] = ORC_DOLLAR; memcpy(tp + 1, text, n); memset(tp + 1 + n, ORC_DOLLAR, (size_t)w);
  /* hash table keyed by the phrase hash (stands in for std::map<uint64_t,word_stats>) */
  uint64_t tsize = 64; while (tsize < 2 * P + 2) tsize <<= 1;
  wstat *tab = calloc(tsize, sizeof *tab);
  uint32_t *slot_of = malloc(P * sizeof *slot_of);     /* phrase -> table slot */
  uint64_t *phash = malloc(P * sizeof *phash);
  uint8_t *last = malloc(P);
  uint64_t *sai = malloc(P * sizeof *sai);
  uint32_t d = 0;
  uint64_t start = 0;                                   /* T' start of the current phrase */
  int rc = 0;
  for (uint64_t k = 0; k < P; k++) {
    uint64_t e = (k < ne) ? ends[k] + 1 : n + (uint64_t)w;   /* T' index of last byte */
    uint64_t len = e - start + 1;
    uint64_t h = orc_kr_hash(tp + start, len);
    phash[k] = h;
    uint64_t s = (h * 0x9E3779B97F4A7C15ULL) & (tsize - 1);
    while (tab[s].occ && tab[s].hash != h) s = (s + 1) & (tsize - 1);
    if (!tab[s].occ) { tab[s].hash = h; tab[s].off = start; tab[s].len = (uint32_t)len; tab[s].occ = 1; d++; }
    else {
      if (tab[s].len != len || memcmp(tp + tab[s].off, tp + start, len)) { rc = -2; break; } /* newscan.cpp:282 */
      tab[s].occ++;
    }
    slot_of[k] = (uint32_t)s;
    last[k] = tp[e - (uint64_t)w];                      /* newscan.cpp:296: w+1 from the end */
    sai[k] = e;                                         /* newscan.cpp:298-299: end position+1 in T */
    start = e - (uint64_t)w + 1;
  }
  if (rc) { free(tab); free(slot_of); free(phash); free(last); free(sai); free(tp); free(ends); return rc; }
  wstat **sorted = malloc((size_t)d * sizeof *sorted);
  uint32_t q = 0;
  for (uint64_t s = 0; s < tsize; s++) if (tab[s].occ) sorted[q++] = &tab[s];
  g_tp = tp;
  qsort(sorted, d, sizeof *sorted, cmp_words);
  uint64_t dsize = 1;
  for (uint32_t i = 0; i < d; i++) dsize += (uint64_t)sorted[i]->len + 1;
  uint8_t *dict = malloc(dsize);
  uint32_t *occ = malloc((size_t)d * sizeof *occ);
  uint64_t o = 0;
  for (uint32_t i = 0; i < d; i++) {
    memcpy(dict + o, tp + sorted[i]->off, sorted[i]->len); o += sorted[i]->len;
    dict[o++] = ORC_ENDOFWORD;
    occ[i] = sorted[i]->occ; sorted[i]->rank = i + 1;   /* 1-based (newscan.cpp:436) */
  }
  dict[o++] = ORC_ENDOFDICT;
  assert(o == dsize);
  uint32_t *parse = malloc(P * sizeof *parse);
  for (uint64_t k = 0; k < P; k++) parse[k] = tab[slot_of[k]].rank;
  out->n_used = n; out->dict = dict; out->dsize = dsize; out->occ = occ; out->d = d;
  out->parse = parse; out->P = P; out->last = last; out->sai = sai; out->phash = phash;
  free(sorted); free(tab); free(slot_of); free(tp); free(ends);
  return 0;
}

/* ------------------------------------------------------------------ suffix sorting */

typedef struct { const uint32_t *rank; uint64_t N, h; } sctx;
static inline uint64_t nextkey(const sctx *c, uint32_t i) {
  uint64_t j = (uint64_t)i + c->h;
  return j < c->N ? (uint64_t)c->rank[j] + 1 : 0;
}
static int cmp_next(const void *a, const void *b, void *cv) {
  const sctx *c = cv;
  uint64_t x = nextkey(c, *(const uint32_t *)a), y = nextkey(c, *(const uint32_t *)b);
  return (x > y) - (x < y);
}

/* Suffix order of an integer string whose keys already carry the symbol order and whose last
 * symbol is the unique minimum.  Prefix doubling: rank[i] = first SA slot of i's group. */
static int suffix_sort(const uint32_t *key, uint64_t N, uint32_t *sa) {
  if (N == 0) return 0;
  if (N >= 0xFFFFFFFEULL) return -1;
  uint32_t *rank = malloc(N * sizeof *rank), *nrank = malloc(N * sizeof *nrank);
  uint32_t *tmp = malloc(N * sizeof *tmp);
  /* LSD radix sort of indices by key, 16-bit digits */
  for (uint64_t i = 0; i < N; i++) sa[i] = (uint32_t)i;
  size_t *cnt = malloc(65537 * sizeof *cnt);
  for (int pass = 0; pass < 2; pass++) {
    int sh = 16 * pass;
    memset(cnt, 0, 65537 * sizeof *cnt);
    for (uint64_t i = 0; i < N; i++) cnt[((key[sa[i]] >> sh) & 0xFFFF) + 1]++;
    for (int b = 0; b < 65536; b++) cnt[b + 1] += cnt[b];
    for (uint64_t i = 0; i < N; i++) tmp[cnt[(key[sa[i]] >> sh) & 0xFFFF]++] = sa[i];
    memcpy(sa, tmp, N * sizeof *sa);
  }
  free(cnt);
  uint64_t unsorted = 0;
  for (uint64_t i = 0; i < N; i++) {
    rank[sa[i]] = (i && key[sa[i]] == key[sa[i - 1]]) ? rank[sa[i - 1]] : (uint32_t)i;
    if (i && key[sa[i]] == key[sa[i - 1]]) unsorted++;
  }
  sctx c = { rank, N, 1 };
  int depth = 0;
  while (unsorted) {
    unsorted = 0; depth++;
    memcpy(nrank, rank, N * sizeof *rank);
    uint64_t i = 0;
    while (i < N) {
      uint64_t j = i + 1;
      while (j < N && rank[sa[j]] == (uint32_t)i) j++;
      if (j - i > 1) {
        qsort_r(sa + i, j - i, sizeof *sa, cmp_next, &c);
        for (uint64_t k = i; k < j; k++) {
          if (k > i && nextkey(&c, sa[k]) == nextkey(&c, sa[k - 1])) { nrank[sa[k]] = nrank[sa[k - 1]]; unsorted++; }
          else nrank[sa[k]] = (uint32_t)k;
        }
      }
      i = j;
    }
    uint32_t *t = rank; rank = nrank; nrank = t; c.rank = rank;
    c.h *= 2;
  }
  free(rank); free(nrank); free(tmp);
  return depth;
}

/* gsacak.c:2492-2495 sacak(): s[n-1] must be 0 */
int orc_sacak(const uint8_t *s, uint32_t *SA, uint64_t n) {
  if (!s || !SA) return -1;
  uint32_t *key = malloc(n * sizeof *key);
  for (uint64_t i = 0; i < n; i++) key[i] = s[i];
  int r = suffix_sort(key, n, SA);
  free(key);
  return r;
}

/* gsacak.c:2497-2500 sacak_int() */
int orc_sacak_int(const uint32_t *s, uint32_t *SA, uint64_t n, uint64_t k) {
  (void)k;
  if (!s || !SA) return -1;
  return suffix_sort(s, n, SA);
}

/* gsacak.c:2502-2522 gsacak(s,SA,LCP,NULL,n): every separator (byte 1) is a distinct symbol
 * ordered by position (gsacak.c:1559-1561), s[n-1]==0; LCP stops at separators
 * (gsa/README.md:76-104 worked example). */
int orc_gsacak(const uint8_t *s, uint32_t *SA, int32_t *LCP, uint64_t n) {
  if (!s || !SA) return -1;
  uint32_t *key = malloc(n * sizeof *key);
  uint32_t nsep = 0;
  for (uint64_t i = 0; i < n; i++) if (s[i] == 1) nsep++;
  uint32_t q = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (s[i] == 0) key[i] = 0;
    else if (s[i] == 1) key[i] = 1 + q++;
    else key[i] = nsep + s[i];
  }
  int r = suffix_sort(key, n, SA);
  free(key);
  if (r < 0) return r;
  if (LCP) {                       /* Kasai over the string with distinct separators */
    uint32_t *isa = malloc(n * sizeof *isa);
    for (uint64_t i = 0; i < n; i++) isa[SA[i]] = (uint32_t)i;
    uint64_t h = 0;
    for (uint64_t i = 0; i < n; i++) {
      uint32_t r0 = isa[i];
      if (r0 == 0) { LCP[0] = 0; h = 0; continue; }
      uint64_t j = SA[r0 - 1];
      while (i + h < n && j + h < n && s[i + h] == s[j + h] && s[i + h] > 1) h++;
      LCP[r0] = (int32_t)h;
      if (h) h--;
    }
    free(isa);
  }
  return r;
}

/* ------------------------------------------------------------------ stage 2: bwtparse */

/* bwtparse.c:212-322 */
int orc_bwtparse(const uint32_t *parse, uint64_t P, const uint8_t *last, const uint64_t *sai,
                 const uint32_t *occ_in, uint32_t d, uint32_t *ilist, uint8_t *bwlast, uint64_t *bwsai) {
  if (P < 2) return -1;                                  /* bwtparse.c:244 assert(n>1) */
  uint64_t n = P;
  uint32_t *Text = malloc((n + 1) * sizeof *Text);
  memcpy(Text, parse, n * sizeof *Text); Text[n] = 0;   /* bwtparse.c:115 */
  uint32_t k = 0;
  for (uint64_t i = 0; i < n; i++) if (Text[i] > k) k = Text[i];
  if (k != d) { free(Text); return -3; }
  uint32_t *SA = malloc((n + 1) * sizeof *SA);
  if (orc_sacak_int(Text, SA, n + 1, (uint64_t)k + 1) < 0) { free(Text); free(SA); return -1; }
  uint32_t *BWT = malloc((n + 1) * sizeof *BWT);
  if (SA[0] != n) { free(Text); free(SA); free(BWT); return -4; }
  BWT[0] = Text[n - 1];                                  /* bwtparse.c:247-249 */
  bwlast[0] = last[n - 2];
  if (sai) bwsai[0] = sai[n - 1];
  for (uint64_t i = 1; i <= n; i++) {
    if (SA[i] == 0) {                                    /* bwtparse.c:252-257 */
      BWT[i] = 0; bwlast[i] = 0; if (sai) bwsai[i] = 0;
    } else {
      bwlast[i] = (SA[i] == 1) ? last[n - 1] : last[SA[i] - 2];   /* bwtparse.c:259-263 */
      if (sai) bwsai[i] = sai[SA[i] - 1];
      BWT[i] = Text[SA[i] - 1];
    }
  }
  /* bwtparse.c:281-303: F = exclusive prefix sums of occ (symbol 0 occurs once) */
  uint32_t *F = malloc(((size_t)k + 1) * sizeof *F);
  F[0] = 0;
  for (uint32_t i = 1; i <= k; i++) F[i] = F[i - 1] + (i == 1 ? 1 : occ_in[i - 2]);
  for (uint64_t i = 0; i <= n; i++) ilist[F[BWT[i]]++] = (uint32_t)i;
  int ok = (ilist[0] == 1) && (BWT[ilist[0]] == 0);     /* bwtparse.c:305-306 */
  free(F); free(BWT); free(SA); free(Text);
  return ok ? 0 : -5;
}

/* ------------------------------------------------------------------ stage 3: pfbwt */

typedef struct { uint8_t *p; uint64_t n, cap; } bbuf;
typedef struct { uint64_t *p; uint64_t n, cap; } qbuf;
static void bpush(bbuf *b, uint8_t c) { if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 1024; b->p = realloc(b->p, b->cap); } b->p[b->n++] = c; }
static void qpush(qbuf *b, uint64_t v) { if (b->n == b->cap) { b->cap = b->cap ? 2 * b->cap : 1024; b->p = realloc(b->p, b->cap * 8); } b->p[b->n++] = v; }

/* pfbwt.cpp:449-473 binsearch + getlen */
static int64_t getlen(uint32_t p, const uint32_t *eos, int64_t n, uint32_t *seqid) {
  int64_t lo = 0, hi = n - 1;
  while (hi > lo) { int64_t mid = (lo + hi) / 2; if (p < eos[mid]) hi = mid; else lo = mid + 1; }
  *seqid = (uint32_t)hi;
  return (int64_t)eos[hi] - (int64_t)p;
}

typedef struct { uint32_t id; uint32_t remaining; const uint32_t *bwtpos; uint8_t ch; } seqid_t;
/* min-heap on *bwtpos (pfbwt.cpp:92-94 inverts '<' to get the same from std::make_heap) */
static void sift_down(seqid_t *h, size_t n, size_t i) {
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && *h[l].bwtpos < *h[m].bwtpos) m = l;
    if (r < n && *h[r].bwtpos < *h[m].bwtpos) m = r;
    if (m == i) return;
    seqid_t t = h[i]; h[i] = h[m]; h[m] = t; i = m;
  }
}

void orc_bwt_free(orc_bwt_t *o) { free(o->bwt); free(o->sa); free(o->ssa); free(o->esa); memset(o, 0, sizeof *o); }

/* pfbwt.cpp:109-242 bwt() with the three writers fwrite_chars_same_suffix{,_sa,_ssa}
 * (pfbwt.cpp:520-676) folded into one emit routine: every emitted char goes through
 * emit(), which applies the run-boundary sampling rules of pfbwt.cpp:163-191 / 614-666. */
typedef struct {
  int flags; bbuf bwt; qbuf sa, ssa, esa;
  int lastbwt; uint64_t lastsa; uint64_t easy, hard;
} emit_t;

static void emit(emit_t *E, int ch, uint64_t sa, int has_sa, int hard) {
  uint64_t pos = E->easy + E->hard;
  if (has_sa && (E->flags & ORC_FLAG_SA)) qpush(&E->sa, sa);
  if (E->flags & (ORC_FLAG_SSA | ORC_FLAG_ESA)) {
    if (pos == 0) {                                           /* pfbwt.cpp:181-190 */
      if (E->flags & ORC_FLAG_SSA) { qpush(&E->ssa, 0); qpush(&E->ssa, sa); }
    } else if (ch != E->lastbwt) {
      if (E->flags & ORC_FLAG_SSA) { qpush(&E->ssa, pos); qpush(&E->ssa, sa); }
      if (E->flags & ORC_FLAG_ESA) { qpush(&E->esa, pos - 1); qpush(&E->esa, E->lastsa); }
    }
    E->lastsa = sa;
  }
  bpush(&E->bwt, (uint8_t)ch);
  E->lastbwt = ch;
  if (hard) E->hard++; else E->easy++;
}

int orc_pfbwt(const uint8_t *dict, uint64_t dsize, const uint32_t *occ, uint32_t dwords,
              const uint32_t *ilist, const uint8_t *bwlast, const uint64_t *bwsai, uint64_t psize,
              int w, int flags, orc_bwt_t *out) {
  memset(out, 0, sizeof *out);
  if ((flags & ORC_FLAG_SA) && (flags & (ORC_FLAG_SSA | ORC_FLAG_ESA))) return -1;  /* bigbwt:59-61 */
  if (flags && !bwsai) return -1;
  if (dsize <= 1 + (uint64_t)w) return -1;                /* pfbwt.cpp:332 */
  uint8_t *d = malloc(dsize); memcpy(d, dict, dsize);
  uint32_t *sa = malloc(dsize * sizeof *sa);
  int32_t *lcp = malloc(dsize * sizeof *lcp);
  if (orc_gsacak(d, sa, lcp, dsize) < 0) { free(d); free(sa); free(lcp); return -1; }
  /* istart: pfbwt.cpp:388-396 */
  uint32_t *istart = malloc(((size_t)dwords + 1) * sizeof *istart);
  uint32_t lastp = 1;
  for (uint32_t i = 0; i < dwords; i++) { istart[i] = lastp; lastp += occ[i]; }
  istart[dwords] = (uint32_t)psize;
  int rc = 0;
  /* sanity checks of pfbwt.cpp:498-512 */
  if (lastp != psize || ilist[0] != 1 || d[0] != ORC_DOLLAR || sa[0] != dsize - 1 ||
      sa[dwords] != dsize - 2 || sa[(uint64_t)dwords + (uint64_t)w + 1] != 0) rc = -6;
  d[0] = 0;                                               /* pfbwt.cpp:126 */
  const uint32_t *eos = sa + 1;                           /* pfbwt.cpp:129 */
  emit_t E; memset(&E, 0, sizeof E);
  E.flags = flags; E.lastbwt = ORC_DOLLAR; E.lastsa = UINT64_MAX;
  uint64_t full_words = 0;
  uint32_t *ids = NULL; uint8_t *chs = NULL; seqid_t *heap = NULL; size_t gcap = 0;
  uint64_t next;
  for (uint64_t i = (uint64_t)dwords + (uint64_t)w + 1; !rc && i < dsize; i = next) {
    next = i + 1;
    uint32_t seqid;
    int64_t suffixLen = getlen(sa[i], eos, dwords, &seqid);
    if (suffixLen <= w) continue;                         /* pfbwt.cpp:151 */
    if (sa[i] == 0 || d[sa[i] - 1] == ORC_ENDOFWORD) {    /* full word, pfbwt.cpp:153-199 */
      full_words++;
      for (uint32_t j = istart[seqid]; j < istart[seqid + 1]; j++) {
        int nextbwt = bwlast[ilist[j]];
        uint64_t sav = 0;
        if (flags) sav = (seqid > 0) ? bwsai[ilist[j]] - (uint64_t)suffixLen : bwsai[0] - (uint64_t)w;
        emit(&E, nextbwt, sav, seqid > 0, 0);
      }
      continue;
    }
    size_t nw = 0;
    if (gcap == 0) { gcap = 16; ids = malloc(gcap * sizeof *ids); chs = malloc(gcap); }
    ids[0] = seqid; chs[0] = d[sa[i] - 1]; nw = 1;
    while (next < dsize && lcp[next] >= suffixLen) {      /* pfbwt.cpp:204-215 */
      uint32_t sid2;
      int64_t l2 = getlen(sa[next], eos, dwords, &sid2);
      if (l2 != suffixLen) break;
      if (nw == gcap) { gcap *= 2; ids = realloc(ids, gcap * sizeof *ids); chs = realloc(chs, gcap); }
      ids[nw] = sid2; chs[nw] = d[sa[next] - 1]; nw++; next++;
    }
    int samechar = 1;
    for (size_t q = 1; q < nw && samechar; q++) samechar = (chs[q - 1] == chs[q]);
    /* BWT only: same char => plain fill (pfbwt.cpp:527-533); with SA info only a single word
     * is "easy" (pfbwt.cpp:568-576, 612-640) */
    if ((!flags && samechar) || nw == 1) {
      for (size_t q = 0; q < nw; q++) {
        uint32_t s = ids[q];
        for (uint32_t j = istart[s]; j < istart[s + 1]; j++)
          emit(&E, chs[q], flags ? bwsai[ilist[j]] - (uint64_t)suffixLen : 0, 1, 0);
      }
    } else {                                              /* heap merge, pfbwt.cpp:537-556 */
      heap = realloc(heap, nw * sizeof *heap);
      for (size_t q = 0; q < nw; q++) {
        uint32_t s = ids[q];
        heap[q].id = s; heap[q].remaining = istart[s + 1] - istart[s];
        heap[q].bwtpos = ilist + istart[s]; heap[q].ch = chs[q];
      }
      size_t hn = nw;
      for (size_t q = hn / 2; q-- > 0;) sift_down(heap, hn, q);
      while (hn) {
        seqid_t *t = &heap[0];
        emit(&E, t->ch, flags ? bwsai[*t->bwtpos] - (uint64_t)suffixLen : 0, 1, 1);
        t->remaining--; t->bwtpos++;
        if (t->remaining == 0) { heap[0] = heap[hn - 1]; hn--; }
        sift_down(heap, hn, 0);
      }
    }
  }
  if (!rc && (flags & ORC_FLAG_ESA)) { qpush(&E.esa, E.easy + E.hard - 1); qpush(&E.esa, E.lastsa); } /* pfbwt.cpp:225-229 */
  if (!rc && full_words != dwords) rc = -7;               /* pfbwt.cpp:230 */
  free(ids); free(chs); free(heap); free(istart); free(lcp); free(sa); free(d);
  out->bwt = E.bwt.p; out->nbwt = E.bwt.n;
  out->sa = E.sa.p; out->nsa = E.sa.n;
  out->ssa = E.ssa.p; out->nssa = E.ssa.n / 2;
  out->esa = E.esa.p; out->nesa = E.esa.n / 2;
  out->full_words = full_words; out->easy = E.easy; out->hard = E.hard;
  if (rc) orc_bwt_free(out);
  return rc;
}

/* ------------------------------------------------------------------ whole chain */

/* bigbwt:69-156 */
int orc_bigbwt(const uint8_t *text, uint64_t n, int w, uint64_t p, int flags, orc_bwt_t *out) {
  orc_parse_t ps;
  int rc = orc_parse(text, n, w, p, &ps);
  if (rc) return rc;
  uint32_t *ilist = malloc((ps.P + 1) * sizeof *ilist);
  uint8_t *bwlast = malloc(ps.P + 1);
  uint64_t *bwsai = malloc((ps.P + 1) * sizeof *bwsai);
  rc = orc_bwtparse(ps.parse, ps.P, ps.last, ps.sai, ps.occ, ps.d, ilist, bwlast, bwsai);
  if (!rc) rc = orc_pfbwt(ps.dict, ps.dsize, ps.occ, ps.d, ilist, bwlast, bwsai, ps.P + 1, w, flags, out);
  free(ilist); free(bwlast); free(bwsai);
  orc_parse_free(&ps);
  return rc;
}

/* simplebwt.c:28-100: SA of text+EOS, BWT[i] = Text[SA[i]-1], EOS where SA[i]==0 */
int orc_simplebwt(const uint8_t *text, uint64_t n, uint8_t *bwt) {
  uint8_t *t = malloc(n + 1);
  memcpy(t, text, n); t[n] = 0;
  uint32_t *SA = malloc((n + 1) * sizeof *SA);
  int rc = orc_sacak(t, SA, n + 1);
  if (rc >= 0) for (uint64_t i = 0; i <= n; i++) bwt[i] = SA[i] ? t[SA[i] - 1] : 0;
  free(SA); free(t);
  return rc < 0 ? rc : 0;
}

/* ------------------------------------------------------------------ synthetic input (SURVEY.md section 4 "GEN") */

/* xorshift64: s^=s<<13; s^=s>>7; s^=s<<17.  Base genome "ACGT"[rnd()&3]; per copy a header
 * ">copy<c>\n", each base replaced by "ACGT"[rnd()&3] when rnd() < (uint64)(r*(2^64-1)) (one
 * draw per base, a second for the replacement), 60 columns + '\n', trailing '\n' on a partial
 * line.  nblocks (start,len) pairs overwrite the base genome with 'N' (BASELINE.md config 2). */
uint64_t orc_gen_fasta(uint64_t G, uint32_t C, double r, uint64_t seed, const uint64_t *nblk, uint32_t nnblk,
                       uint8_t *out, uint64_t cap) {
  uint64_t s = seed, o = 0;
#define RND() (s ^= s << 13, s ^= s >> 7, s ^= s << 17, s)
  uint8_t *base = malloc(G ? G : 1);
  for (uint64_t i = 0; i < G; i++) base[i] = (uint8_t)"ACGT"[RND() & 3];
  for (uint32_t b = 0; b < nnblk; b++)
    for (uint64_t i = nblk[2 * b]; i < nblk[2 * b] + nblk[2 * b + 1] && i < G; i++) base[i] = 'N';
  uint64_t thr = (uint64_t)(r * 18446744073709551615.0);
  for (uint32_t c = 0; c < C; c++) {
    char hdr[32];
    int hl = snprintf(hdr, sizeof hdr, ">copy%u\n", c);
    if (o + (uint64_t)hl > cap) { free(base); return 0; }
    memcpy(out + o, hdr, (size_t)hl); o += (uint64_t)hl;
    for (uint64_t i = 0; i < G; i++) {
      uint8_t ch = base[i];
      if (r > 0 && RND() < thr) ch = (uint8_t)"ACGT"[RND() & 3];
      if (o + 2 > cap) { free(base); return 0; }
      out[o++] = ch;
      if (i % 60 == 59 || i == G - 1) out[o++] = '\n';
    }
  }
#undef RND
  free(base);
  return o;
}

/* ---- FASTA/FASTQ input (bigbwt -f) -------------------------------------------------------------
 * Restates kseq.h's reader as newscan.cpp:332-352 drives it.  rd_getc = ks_getc (kseq.h:77-91),
 * rd_line = ks_getuntil2(KS_SEP_LINE, append=1) (kseq.h:94-146): appends the rest of the line,
 * removes a trailing '\r' when the string is longer than one char, returns -1 only when no byte
 * was available.  rd_word = ks_getuntil2(KS_SEP_SPACE) and reports the delimiter. */
typedef struct { const uint8_t *b; uint64_t n, i; } rd_t;
typedef struct { uint8_t *s; uint64_t l; } str_t;
static int rd_getc(rd_t *r) { return r->i < r->n ? r->b[r->i++] : -1; }
static int rd_line(rd_t *r, str_t *s, int keep) {
  if (r->i >= r->n) return -1;
  while (r->i < r->n && r->b[r->i] != '\n') { if (keep) s->s[s->l] = r->b[r->i]; s->l++; r->i++; }
  if (r->i < r->n) r->i++;
  if (keep) { if (s->l > 1 && s->s[s->l - 1] == '\r') s->l--; }
  return 0;
}
static int rd_word(rd_t *r, int *delim) {
  *delim = 0;
  if (r->i >= r->n) return -1;
  while (r->i < r->n && !isspace(r->b[r->i])) r->i++;
  if (r->i < r->n) *delim = r->b[r->i++];
  return 0;
}
/* one kseq_read (kseq.h:178-222): >=0 sequence length in seq, -1 end of file, -2 truncated quality */
static int64_t fasta_next(rd_t *r, int *last_char, str_t *seq, uint8_t *scratch) {
  int c;
  if (*last_char == 0) {
    while ((c = rd_getc(r)) >= 0 && c != '>' && c != '@') {}
    if (c < 0) return -1;
    *last_char = c;
  }
  seq->l = 0;
  if (rd_word(r, &c) < 0) return -1;
  if (c != '\n') { str_t dump = {NULL, 0}; rd_line(r, &dump, 0); }
  while ((c = rd_getc(r)) >= 0 && c != '>' && c != '+' && c != '@') {
    if (c == '\n') continue;
    seq->s[seq->l++] = (uint8_t)c;
    rd_line(r, seq, 1);
  }
  if (c == '>' || c == '@') *last_char = c;
  if (c != '+') return (int64_t)seq->l;
  while ((c = rd_getc(r)) >= 0 && c != '\n') {}
  if (c < 0) return -2;
  str_t q = {scratch, 0};
  while (rd_line(r, &q, 1) >= 0 && q.l < seq->l) {}
  *last_char = 0;
  if (seq->l != q.l) return -2;
  return (int64_t)seq->l;
}
uint64_t orc_fasta_text(const uint8_t *in, uint64_t n, uint8_t *out) {
  rd_t r = {in, n, 0};
  uint8_t *buf = malloc(n ? n : 1), *scratch = malloc(n ? n : 1);
  str_t seq = {buf, 0};
  int last_char = 0;
  uint64_t o = 0;
  int64_t l;
  int stop = 0;
  while (!stop && (l = fasta_next(&r, &last_char, &seq, scratch)) >= 0) {
    for (int64_t i = 0; i < l; i++) {
      int ch = toupper(seq.s[i]);
      if (ch <= 2) { stop = 1; break; }      /* newscan.cpp:340-343, 349 */
      out[o++] = (uint8_t)ch;
    }
  }
  free(buf); free(scratch);
  return o;
}
