/* stages.c -- the reference's three stage executables and `unparse` with their own command lines
 * and file formats, each one a single call into libpfpgpu.so.  The program acts by the name it is
 * run under (the Makefile installs it as bin/newscanNT.x, newscan.x, pscan.x, bwtparse,
 * bwtparse64, pfbwtNT.x, pfbwt.x, pfbwtNT64.x, pfbwt64.x, simplebwt, simplebwt64, unparse), so the reference's `bigbwt`
 * script works unchanged when its directory holds these, and any one stage can be swapped with
 * the reference's and diffed file by file.
 *
 *   newscan[NT].x | pscan.x  [-w W] [-p M] [-s] [-c] [-f] [-P] [-v] [-t T] file
 *        newscan.cpp:476-560 -> file.dict|.dicz .occ .parse .last [.sai]
 *        (-t T > 0: .last/.sai are written as T segments file.<i>.last, as the reference's threaded
 *        scanners do, because `bwtparse -t T` reads them that way, utils.c:57-105)
 *   bwtparse[64]  file [-s] [-t T]      bwtparse.c:140-160 -> file.ilist .bwlast [.bwsai]
 *   pfbwt[NT][64].x  [-w W] [-s] [-e] [-S] [-t T] file     pfbwt.cpp:255-320 -> file.bwt [.sa|.ssa|.esa]
 *   simplebwt[64] file                  simplebwt.c:28-100: whole-text suffix array -> file.Bwt
 *   unparse [-o out] file               unparse.c:76-140: file.dicz + file.parse -> file.out
 */
#define _GNU_SOURCE
#include <errno.h>
#include <libgen.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "fasta.h"
#include "pfpgpu.h"

static void die(const char *msg) {   /* utils.c:12-16 */
  if (errno) perror(msg); else fprintf(stderr, "%s\n", msg);
  exit(1);
}

static void show_command_line(int argc, char **argv) {
  puts("==== Command line:");
  for (int i = 0; i < argc; i++) printf(" %s", argv[i]);
  puts("");
}

static void put_file(const char *base, const char *ext, const void *data, size_t bytes) {
  char name[4096];
  snprintf(name, sizeof name, "%s.%s", base, ext);
  FILE *f = fopen(name, "wb");
  if (!f) die(name);
  if (bytes && fwrite(data, 1, bytes, f) != bytes) die(name);
  if (fclose(f) != 0) die(name);
}

/* whole file -> malloc'ed buffer (+16 spare bytes); missing file is fatal unless optional */
static uint8_t *get_file(const char *base, const char *ext, size_t *bytes, int optional) {
  char name[4096];
  if (ext) snprintf(name, sizeof name, "%s.%s", base, ext); else snprintf(name, sizeof name, "%s", base);
  FILE *f = fopen(name, "rb");
  if (!f) { if (optional) return NULL; die(name); }
  if (fseek(f, 0, SEEK_END) != 0) die(name);
  long sz = ftell(f);
  rewind(f);
  uint8_t *b = malloc((size_t)sz + 16);
  if (!b) die("Out of memory");
  if (sz && fread(b, 1, (size_t)sz, f) != (size_t)sz) die(name);
  fclose(f);
  *bytes = (size_t)sz;
  return b;
}

/* a file that may be split into segments base.<i>.ext, i < nsegs (utils.c:57-105); nsegs 0 = one file */
static uint8_t *get_segmented(const char *base, const char *ext, int nsegs, size_t *bytes) {
  if (nsegs == 0) return get_file(base, ext, bytes, 0);
  uint8_t *all = malloc(16);
  size_t len = 0;
  for (int i = 0; i < nsegs; i++) {
    char e2[64];
    size_t b = 0;
    snprintf(e2, sizeof e2, "%d.%s", i, ext);
    uint8_t *part = get_file(base, e2, &b, 0);
    all = realloc(all, len + b + 16);
    if (!all) die("Out of memory");
    memcpy(all + len, part, b);
    len += b;
    free(part);
  }
  *bytes = len;
  return all;
}

static void put_segmented(const char *base, const char *ext, int nsegs, const uint8_t *data, size_t items, size_t item_bytes) {
  if (nsegs == 0) { put_file(base, ext, data, items * item_bytes); return; }
  for (int i = 0; i < nsegs; i++) {
    char e2[64];
    size_t lo = items * (size_t)i / (size_t)nsegs, hi = items * (size_t)(i + 1) / (size_t)nsegs;
    snprintf(e2, sizeof e2, "%d.%s", i, ext);
    put_file(base, e2, data + lo * item_bytes, (hi - lo) * item_bytes);
  }
}

static pfp_ctx *open_gpu(void) {
  pfp_ctx *ctx = NULL;
  int rc = pfp_ctx_create(&ctx, getenv("PFP_DEVICE") ? atoi(getenv("PFP_DEVICE")) : 0);
  if (rc) { fprintf(stderr, "Cannot initialise the GPU (%s): this tool has no CPU path\n", pfp_strerror(rc)); exit(1); }
  return ctx;
}

static void fail(pfp_ctx *ctx, int rc) {
  fprintf(stderr, "%s: %s\n", pfp_strerror(rc), pfp_last_error(ctx));
  exit(1);
}

/* ---------------------------------------------------------------------------------- newscan */
static int main_newscan(int argc, char **argv, int nothreads) {
  int w = 10, p = 100, sa = 0, compress = 0, fasta = 0, th = 0, verbose = 0, c;
  time_t start = time(NULL);
  show_command_line(argc, argv);
  while ((c = getopt(argc, argv, "p:w:fsPcht:v")) != -1) {
    switch (c) {
      case 's': sa = 1; break;
      case 'P': break;                       /* deduplication here is exact */
      case 'c': compress = 1; break;
      case 'w': w = atoi(optarg); break;
      case 'p': p = atoi(optarg); break;
      case 'f': fasta = 1; break;
      case 't': th = atoi(optarg); break;
      case 'v': verbose++; break;
      default:
        printf("Usage: %s <input filename> [options]\n  -w W window size (10)  -p M modulus (100)  -t T segments for .last/.sai\n"
               "  -s suffix array info  -c compress the output dictionary (.dicz)  -f fasta/fastq input  -v verbose\n", argv[0]);
        exit(1);
    }
  }
  if (argc != optind + 1) { puts("Invalid number of arguments"); exit(1); }
  const char *name = argv[optind];
  if (w < 4) { puts("Windows size must be at least 4"); exit(1); }       /* newscan.cpp:537-544 */
  if (p < 10) { puts("Modulus must be at least 10"); exit(1); }
  if (nothreads && th != 0) { puts("The NT version cannot use threads"); exit(1); }
  if (th < 0) { puts("Number of threads cannot be negative"); exit(1); }
  printf("Windows size: %d\nStop word modulus: %d\n", w, p);
  size_t n = 0;
  uint8_t *text;
  if (fasta) {
    size_t raw_n = 0;
    uint8_t *raw = pfp_read_maybe_gz(name, &raw_n);
    if (!raw) die(name);
    text = malloc(raw_n + 16);
    if (!text) die("Out of memory");
    n = pfp_fasta_text(raw, raw_n, text);
    free(raw);
  } else {
    text = get_file(name, NULL, &n, 0);
  }
  pfp_ctx *ctx = open_gpu();
  pfp_parse_result r;
  int rc = pfp_parse(ctx, text, n, w, (uint64_t)p, sa, &r);
  if (rc) fail(ctx, rc);
  printf("Found %llu distinct words\nParsing took: %ld wall clock seconds\n", (unsigned long long)r.n_words, (long)(time(NULL) - start));
  if (verbose) printf("Total number of words: %llu\nSum of lenghts of dictionary words: %llu\n", (unsigned long long)r.n_phrases,
                      (unsigned long long)(r.dict_size - r.n_words - 1));
  if (compress) {
    uint8_t *z = malloc(r.dict_size + 1);
    if (!z) die("Out of memory");
    size_t o = pfp_dicz_from_dict(r.dict, r.dict_size, w, z);
    put_file(name, "dicz", z, o);
    free(z);
  } else {
    put_file(name, "dict", r.dict, r.dict_size);
  }
  put_file(name, "occ", r.occ, r.n_words * 4);
  put_file(name, "parse", r.parse, r.n_phrases * 4);
  put_segmented(name, "last", th, r.last, r.n_phrases, 1);
  if (sa) put_segmented(name, "sai", th, r.sai, r.n_phrases, 5);
  pfp_parse_result_free(&r);
  pfp_ctx_destroy(ctx);
  printf("==== Elapsed time: %ld wall clock seconds\n", (long)(time(NULL) - start));
  return 0;
}

/* --------------------------------------------------------------------------------- bwtparse */
static int main_bwtparse(int argc, char **argv) {
  int sa = 0, th = 0, c;
  time_t start = time(NULL);
  show_command_line(argc, argv);
  puts("");
  while ((c = getopt(argc, argv, "sht:")) != -1) {
    switch (c) {
      case 's': sa = 1; break;
      case 't': th = atoi(optarg); break;
      default:
        printf("Usage: %s <basename> [options]\n\nCompute the BWT of basename.parse and store its inverted list occurrence\n"
               "Permute the file basename.last according to the same permutation\n  -s  permute also sa info\n"
               "  -t T  .last/.sai are split in T segments\n", argv[0]);
        exit(1);
    }
  }
  if (argc != optind + 1) { printf("Usage: %s <basename> [options]\n", argv[0]); exit(1); }
  const char *base = argv[optind];
  size_t pb = 0, lb = 0, sb = 0, ob = 0;
  uint8_t *parse = get_file(base, "parse", &pb, 0);
  if (pb % 4 != 0) { puts("Invalid input file: size not multiple of 4"); exit(1); }   /* bwtparse.c:81-84 */
  size_t P = pb / 4;
  printf("Parse file contains %zu words\n", P);
  uint8_t *last = get_segmented(base, "last", th, &lb);
  uint8_t *sai = sa ? get_segmented(base, "sai", th, &sb) : NULL;
  uint8_t *occ = get_file(base, "occ", &ob, 0);
  if (lb != P) die("Error reading the .last file: wrong size");
  if (sa && sb != 5 * P) die("Error reading the .sai file: wrong size");
  pfp_ctx *ctx = open_gpu();
  uint32_t *ilist = malloc((P + 1) * 4);
  uint8_t *bwlast = malloc(P + 1), *bwsai = sa ? malloc((P + 1) * 5) : NULL;
  if (!ilist || !bwlast || (sa && !bwsai)) die("Out of memory");
  int rc = pfp_bwtparse(ctx, (const uint32_t *)parse, P, last, sai, (const uint32_t *)occ, ob / 4, ilist, bwlast, bwsai);
  if (rc) fail(ctx, rc);
  put_file(base, "ilist", ilist, (P + 1) * 4);
  put_file(base, "bwlast", bwlast, P + 1);
  if (sa) put_file(base, "bwsai", bwsai, (P + 1) * 5);
  pfp_ctx_destroy(ctx);
  free(parse); free(last); free(sai); free(occ); free(ilist); free(bwlast); free(bwsai);
  printf("==== Elapsed time: %ld wall clock seconds\n", (long)(time(NULL) - start));
  return 0;
}

/* ------------------------------------------------------------------------------------ pfbwt */
static int main_pfbwt(int argc, char **argv, int nothreads) {
  int w = 10, s = 0, e = 0, S = 0, th = 0, c;
  time_t start = time(NULL);
  show_command_line(argc, argv);
  while ((c = getopt(argc, argv, "t:w:sehS")) != -1) {
    switch (c) {
      case 's': s = 1; break;
      case 'e': e = 1; break;
      case 'S': S = 1; break;
      case 'w': w = atoi(optarg); break;
      case 't': th = atoi(optarg); break;
      default:
        printf("Usage: %s <input filename> [options]\n  -w W window size (10)  -s/-e sampled SA at run starts/ends  -S full SA\n", argv[0]);
        exit(1);
    }
  }
  if (argc != optind + 1) { puts("Invalid number of arguments"); exit(1); }
  const char *base = argv[optind];
  if (S && (s || e)) { printf("You can either require the sampled SA or the full SA, not both"); exit(1); }   /* pfbwt.cpp:298-301 */
  if (w < 4) { puts("Windows size must be at least 4"); exit(1); }
  if (nothreads && th != 0) { puts("The NT version cannot use threads"); exit(1); }
  int flags = (S ? PFP_FLAG_SA : 0) | (s ? PFP_FLAG_SSA : 0) | (e ? PFP_FLAG_ESA : 0);
  size_t db = 0, ob = 0, ib = 0, lb = 0, sb = 0;
  uint8_t *dict = get_file(base, "dict", &db, 0);
  uint8_t *occ = get_file(base, "occ", &ob, 0);
  uint8_t *ilist = get_file(base, "ilist", &ib, 0);
  uint8_t *bwlast = get_file(base, "bwlast", &lb, 0);
  uint8_t *bwsai = flags ? get_file(base, "bwsai", &sb, 0) : NULL;
  if (ib % 4 != 0 || lb != ib / 4) die("Invalid ilist/bwlast files");                /* pfbwt.cpp:350-372 */
  if (flags && sb != 5 * lb) die("Invalid bwsai file");
  printf("Dictionary file size: %zu\nDictionary words: %zu\nParse size: %zu\n", db, ob / 4, lb);
  pfp_ctx *ctx = open_gpu();
  pfp_bwt_result br;
  int rc = pfp_merge(ctx, dict, db, (const uint32_t *)occ, ob / 4, (const uint32_t *)ilist, bwlast, bwsai, lb, w, flags, &br);
  if (rc) fail(ctx, rc);
  put_file(base, "bwt", br.bwt, br.bwt_size);
  if (S) put_file(base, "sa", br.sa, br.sa_bytes);
  if (s) put_file(base, "ssa", br.ssa, br.ssa_bytes);
  if (e) put_file(base, "esa", br.esa, br.esa_bytes);
  pfp_bwt_result_free(&br);
  pfp_ctx_destroy(ctx);
  free(dict); free(occ); free(ilist); free(bwlast); free(bwsai);
  printf("==== Elapsed time: %ld wall clock seconds\n", (long)(time(NULL) - start));
  return 0;
}

/* -------------------------------------------------------------------------------- simplebwt */
static int main_simplebwt(int argc, char **argv) {
  time_t start = time(NULL);
  show_command_line(argc, argv);
  if (argc != 2) { printf("Usage: %s <input filename>\n", argv[0]); exit(1); }
  size_t n = 0;
  uint8_t *text = get_file(argv[1], NULL, &n, 0);
  if (n + 1 >= 0xFFFFFFF0ull) die("simplebwt: input too large for 32-bit suffix array entries");
  text[n] = 0;                                         /* simplebwt.c:62: 0 is the end-of-string symbol */
  for (size_t i = 0; i < n; i++) if (text[i] == 0) die("Input file contains a 0 byte");
  uint32_t *SA = malloc((n + 1) * sizeof *SA);
  uint8_t *B = malloc(n + 1);
  if (!SA || !B) die("Out of memory");
  pfp_ctx *ctx = open_gpu();
  int rc = pfp_sacak(ctx, text, SA, n + 1);
  if (rc) fail(ctx, rc);
  for (size_t i = 0; i <= n; i++) B[i] = SA[i] ? text[SA[i] - 1] : 0;      /* simplebwt.c:80-93 */
  put_file(argv[1], "Bwt", B, n + 1);
  pfp_ctx_destroy(ctx);
  free(text); free(SA); free(B);
  printf("==== Elapsed time: %ld wall clock seconds\n", (long)(time(NULL) - start));
  return 0;
}

/* ---------------------------------------------------------------------------------- unparse */
static int main_unparse(int argc, char **argv) {
  char *outname = NULL;
  int c;
  time_t start = time(NULL);
  while ((c = getopt(argc, argv, "ho:")) != -1) {
    switch (c) {
      case 'o': outname = strdup(optarg); break;
      default: printf("Usage: %s <basename> [-o outfile]\nRebuild the text from basename.dicz and basename.parse\n", argv[0]); exit(1);
    }
  }
  if (argc != optind + 1) { printf("Usage: %s <basename> [-o outfile]\n", argv[0]); exit(1); }
  const char *base = argv[optind];
  if (!outname && asprintf(&outname, "%s.out", base) < 0) die("Error creating output file name");
  size_t db = 0, pb = 0;
  uint8_t *dicz = get_file(base, "dicz", &db, 0);
  uint8_t *parse = get_file(base, "parse", &pb, 0);
  size_t words = 0, cap = 1024;
  size_t *wstart = malloc(cap * sizeof *wstart);
  wstart[0] = 0;
  for (size_t i = 0; i < db; i++) {
    if (dicz[i] != 1) continue;
    if (++words == cap) { cap *= 2; wstart = realloc(wstart, cap * sizeof *wstart); if (!wstart) die("Allocation error"); }
    wstart[words] = i + 1;
  }
  fprintf(stderr, "Found %zu dictionary words\nRecovering file %s\n", words, outname);
  FILE *f = fopen(outname, "wb");
  if (!f) die("Cannot open output file");
  for (size_t k = 0; k < pb / 4; k++) {
    uint32_t id;
    memcpy(&id, parse + 4 * k, 4);
    if (id == 0 || id - 1 >= words) die("Invalid word ID in the parse file");       /* unparse.c:124 */
    size_t b = wstart[id - 1], e = wstart[id] - 1;
    if (e > b && fwrite(dicz + b, 1, e - b, f) != e - b) die("Error writing to the output file");
  }
  if (fclose(f) != 0) die("Error writing to the output file");
  free(wstart); free(dicz); free(parse); free(outname);
  printf("==== Elapsed time: %ld wall clock seconds\n", (long)(time(NULL) - start));
  return 0;
}

int main(int argc, char **argv) {
  char *self = strdup(argv[0]);
  const char *me = basename(self);
  if (!strncmp(me, "newscanNT", 9)) return main_newscan(argc, argv, 1);
  if (!strncmp(me, "newscan", 7) || !strncmp(me, "pscan", 5)) return main_newscan(argc, argv, 0);
  if (!strncmp(me, "bwtparse", 8)) return main_bwtparse(argc, argv);
  if (!strncmp(me, "pfbwtNT", 7)) return main_pfbwt(argc, argv, 1);
  if (!strncmp(me, "pfbwt", 5)) return main_pfbwt(argc, argv, 0);
  if (!strncmp(me, "simplebwt", 9)) return main_simplebwt(argc, argv);
  if (!strncmp(me, "unparse", 7)) return main_unparse(argc, argv);
  fprintf(stderr, "%s: run me as newscan[NT].x, pscan.x, bwtparse[64], pfbwt[NT][64].x, simplebwt[64] or unparse\n", me);
  return 2;
}
