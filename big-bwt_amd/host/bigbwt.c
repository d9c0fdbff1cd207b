/* bigbwt.c -- C command-line driver with the flag surface of the reference's `bigbwt` script
 * (reference bigbwt:36-197), calling the HIP kernels in-process through the C ABI
 * (include/pfpgpu.h).  Where the reference chains three executables through temp files
 * (newscan -> bwtparse -> pfbwt, bigbwt:69-156) this driver makes one pfp_bigbwt() call and
 * every intermediate stays in HBM; -k additionally materialises the reference's temp files
 * through the staged entry points so that each one can be diffed against the reference.
 *
 *   bigbwt [-w W] [-p M] [-t T] [-s] [-e] [-S] [-k] [-v] [-c] [-f] [--sum] [--parsing]
 *          [--compress] [-P] input
 *
 * Outputs next to the input, byte formats as the reference: .bwt always; .sa | .ssa | .esa;
 * .log (appended).  -t is accepted and ignored (the GPU is the helper); -P is accepted and
 * unnecessary (phrase deduplication is exact).  -f reads FASTA/FASTQ (plain or gzip) exactly as the
 * reference's kseq reader does (fasta.c); with -f the -c check runs on that filtered text (the
 * reference compares against the raw file, bigbwt:183-184, which can never match).
 * --parsing and --compress write the overlap-free dictionary (.dicz) as the reference does
 * (bigbwt:81); --compress then shells out to tar/xz exactly as bigbwt:98 does.
 * -G N (--gpus N) builds ONE BWT on N GPUs of the node: one call, pfp_bigbwt_files_multi - a host thread and a context
 * per GPU inside the library, RCCL collectives between them, every rank reads its byte range of the text and pwrites its
 * ranges of the output files (the reference's -t N threads do the same on the CPU); --sum and -c then run as for one GPU.
 * PFP_MULTI_PYTHON=1 starts N processes of ../dist_main.py under torch.distributed instead (the driver the bench uses).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <getopt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include "pfpgpu.h"
#include "fasta.h"

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int write_file(const char *base, const char *ext, const void *data, size_t bytes) {
  char name[4096];
  snprintf(name, sizeof name, "%s.%s", base, ext);
  FILE *f = fopen(name, "wb");
  if (!f) { perror(name); return -1; }
  size_t w = bytes ? fwrite(data, 1, bytes, f) : 0;
  if (fclose(f) != 0 || w != bytes) { fprintf(stderr, "Error writing %s\n", name); return -1; }
  return 0;
}

/* bigbwt:219-228 file_digest: shells out to sha256sum exactly as the reference does */
static void print_digest(const char *label, const char *base, const char *ext) {
  char cmd[4200], line[256] = "Error!";
  snprintf(cmd, sizeof cmd, "sha256sum '%s.%s' 2>/dev/null", base, ext);
  FILE *p = popen(cmd, "r");
  if (p) {
    if (fgets(line, sizeof line, p)) { char *sp = strchr(line, ' '); if (sp) *sp = 0; }
    pclose(p);
  }
  printf("%s sha256sum: %s\n", label, line);
}

/* a TCP port nobody listens on right now (the rendezvous of the ranks, on 127.0.0.1) */
static int free_port(void) {
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) return 29500;
  struct sockaddr_in a;
  memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  socklen_t len = sizeof a;
  int port = 29500;
  if (bind(fd, (struct sockaddr *)&a, sizeof a) == 0 && getsockname(fd, (struct sockaddr *)&a, &len) == 0) port = ntohs(a.sin_port);
  close(fd);
  return port;
}

/* -G N: N ranks of dist_main.py (next to this executable) under torch.distributed.run, as a child process; this
 * process has not initialised the GPU yet.  Returns the launcher's exit code (non-zero if any rank failed). */
static int run_ranks(int gpus, const char *textfile, const char *base, int w, unsigned long long p, int flags,
                     unsigned long long halo, int verbose) {
  char exe[4096], script[4200], nproc[64], port[32], ws[32], ps[32], fs[32], hs[32];
  ssize_t len = readlink("/proc/self/exe", exe, sizeof exe - 1);
  if (len <= 0) { perror("/proc/self/exe"); return 127; }
  exe[len] = 0;
  char *slash = strrchr(exe, '/');
  if (slash) *slash = 0;
  snprintf(script, sizeof script, "%s/dist_main.py", exe);
  if (access(script, R_OK) != 0) { perror(script); return 127; }
  const char *py = getenv("PFP_PYTHON");
  if (!py || !*py) py = "python3";
  snprintf(nproc, sizeof nproc, "--nproc-per-node=%d", gpus);
  snprintf(port, sizeof port, "%d", free_port());
  snprintf(ws, sizeof ws, "%d", w);
  snprintf(ps, sizeof ps, "%llu", p);
  snprintf(fs, sizeof fs, "%d", flags);
  snprintf(hs, sizeof hs, "%llu", halo);
  const char *args[32];
  int k = 0;
  args[k++] = py; args[k++] = "-m"; args[k++] = "torch.distributed.run"; args[k++] = "--nnodes=1"; args[k++] = nproc;
  args[k++] = "--master-addr"; args[k++] = "127.0.0.1"; args[k++] = "--master-port"; args[k++] = port;
  args[k++] = script; args[k++] = "--text"; args[k++] = textfile; args[k++] = "--base"; args[k++] = base;
  args[k++] = "-w"; args[k++] = ws; args[k++] = "-p"; args[k++] = ps; args[k++] = "--flags"; args[k++] = fs;
  args[k++] = "--halo"; args[k++] = hs;
  if (verbose) args[k++] = "-v";
  args[k] = NULL;
  fflush(NULL);
  pid_t pid = fork();
  if (pid < 0) { perror("fork"); return 127; }
  if (pid == 0) {
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);   /* dmabuf IPC: what RCCL needs between processes on this driver */
    setenv("OMP_NUM_THREADS", "4", 0);
    execvp(py, (char *const *)args);
    perror(py);
    _exit(127);
  }
  int st = 0;
  while (waitpid(pid, &st, 0) < 0)
    if (errno != EINTR) { perror("waitpid"); return 127; }
  return WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
}

static void usage(const char *argv0) {
  printf("usage: %s [-h] [-w WSIZE] [-p MOD] [-t T] [-s] [-e] [-S] [-k] [-v] [-c] [-f] [--sum]\n"
         "              [--parsing] [--compress] [--probing] [-G N] input\n\n"
         "MI355X build of the prefix-free-parsing BWT tool (drop-in for alshai/Big-BWT's bigbwt).\n\n"
         "  input            input file name\n"
         "  -w, --wsize W    sliding window size (def. 10)\n"
         "  -p, --mod M      hash modulus (def. 100)\n"
         "  -t T             number of helper threads (accepted, ignored: the GPU does the work)\n"
         "  -s               compute the start run-length sampled Suffix Array\n"
         "  -e               compute the end run-length sampled Suffix Array\n"
         "  -S               compute the full Suffix Array\n"
         "  -k               keep temporary files (.dict .occ .parse .last .sai .ilist .bwlast .bwsai)\n"
         "  -v               verbose\n"
         "  -c               check BWT against the whole-text suffix array (reference: SACA-K)\n"
         "  -f               read fasta/fastq, plain or gzip (headers and newlines dropped, upper-cased)\n"
         "  --sum            compute output files sha256sum\n"
         "  --parsing        stop after the parsing phase (debug only)\n"
         "  --compress       compress output of the parsing phase (.parse.txz of .parse and .dicz)\n"
         "  -P, --probing    accepted for compatibility (deduplication here is exact)\n"
         "  -G, --gpus N     one BWT on N GPUs of this node, devices --device .. --device + N - 1 (0: all; RCCL; not with -k / --parsing / --compress)\n"
         "      --density D  single GPU without -k: cut the text with probability D / MOD instead of 1 / MOD (same outputs; shorter\n"
         "                   phrases - D = 2 - suit collections of many near-identical copies; def. 1)\n"
         "      --halo H     -G: bytes of each byte range its right neighbour also reads (def. 1048576; must cover a phrase)\n",
         argv0);
}

int main(int argc, char **argv) {
  const double t_main = now_s();
  int w = 10, th = 0, s = 0, e = 0, S = 0, keep = 0, verbose = 0, check = 0, fasta = 0, sum = 0, parsing = 0,
      compress = 0, device = 0, gpus = 1;
  unsigned long long p = 100, halo = 1ull << 20;
  double density = 0;
  static struct option lo[] = {{"wsize", required_argument, 0, 'w'}, {"mod", required_argument, 0, 'p'},
                               {"sum", no_argument, 0, 1000},        {"parsing", no_argument, 0, 1001},
                               {"compress", no_argument, 0, 1002},   {"probing", no_argument, 0, 'P'},
                               {"device", required_argument, 0, 1003}, {"help", no_argument, 0, 'h'},
                               {"gpus", required_argument, 0, 'G'},  {"halo", required_argument, 0, 1004},
                               {"density", required_argument, 0, 1005},
                               {0, 0, 0, 0}};
  int c;
  while ((c = getopt_long(argc, argv, "w:p:t:seSkvcfPhG:", lo, NULL)) != -1) {
    switch (c) {
      case 'w': w = atoi(optarg); break;
      case 'p': p = strtoull(optarg, NULL, 10); break;
      case 't': th = atoi(optarg); break;
      case 's': s = 1; break;
      case 'e': e = 1; break;
      case 'S': S = 1; break;
      case 'k': keep = 1; break;
      case 'v': verbose = 1; break;
      case 'c': check = 1; break;
      case 'f': fasta = 1; break;
      case 'P': break;
      case 1000: sum = 1; break;
      case 1001: parsing = 1; break;
      case 1002: compress = 1; break;
      case 1003: device = atoi(optarg); break;
      case 1004: halo = strtoull(optarg, NULL, 10); break;
      case 1005: density = atof(optarg); break;
      case 'G': gpus = atoi(optarg); break;
      case 'h': usage(argv[0]); return 0;
      default: usage(argv[0]); return 2;
    }
  }
  if (optind + 1 != argc) { usage(argv[0]); return 2; }
  const char *input = argv[optind];
  (void)th;
  if (S && (s || e)) {   /* bigbwt:59-61 */
    printf("You can either compute the full SA or a sample of it, not both. Exiting...\n");
    return 0;
  }
  if (gpus == 0) {   /* -G 0: every GPU this process can see */
    /* (the Python ranks are started by fork + exec: this process must not have touched HIP before that - pfp_device_count
     *  initialises it - so with PFP_MULTI_PYTHON the count has to be given) */
    const char *py = getenv("PFP_MULTI_PYTHON");
    if (py && atoi(py) != 0) {
      fprintf(stderr, "-G 0 counts the devices through HIP, which the launcher of the Python ranks (PFP_MULTI_PYTHON=1) must not have initialised: pass -G N\n");
      return 2;
    }
    gpus = pfp_device_count() - device;
    if (gpus < 1) { fprintf(stderr, "Cannot initialise the GPU (no device %d): this tool has no CPU path\n", device); return 1; }
  }
  if (gpus < 1 || (gpus > 1 && (keep || parsing || compress))) {
    printf("-G N needs N >= 1 and writes no temporary files: not with -k, --parsing or --compress. Exiting...\n");
    return 2;
  }
  char logname[4096];
  snprintf(logname, sizeof logname, "%s.log", input);
  printf("Sending logging messages to file: %s\n", logname);
  FILE *logf = fopen(logname, "a");
  if (!logf) { perror(logname); return 1; }

  uint64_t n = 0;
  const uint8_t *text = NULL;
  int fd_in = -1;
  if (fasta) {      /* newscan.cpp:332-352: gzopen + kseq, sequences only */
    size_t raw_n = 0;
    uint8_t *raw = pfp_read_maybe_gz(input, &raw_n);
    if (!raw) { perror(input); return 1; }
    uint8_t *seq = malloc(raw_n ? raw_n : 1);
    if (!seq) { fprintf(stderr, "out of memory\n"); return 1; }
    n = pfp_fasta_text(raw, raw_n, seq);
    free(raw);
    text = seq;
  } else {
    fd_in = open(input, O_RDONLY);
    struct stat sb;
    if (fd_in < 0 || fstat(fd_in, &sb) != 0) { perror(input); return 1; }
    n = (uint64_t)sb.st_size;
    /* (the mapping is what the staged entry points, -G N and -c read; the plain single-GPU run reads the file with pread) */
    text = n ? mmap(NULL, n, PROT_READ, MAP_PRIVATE, fd_in, 0) : (const uint8_t *)"";
    if (text == MAP_FAILED) { perror("mmap"); return 1; }
  }

  int flags = (S ? PFP_FLAG_SA : 0) | (s ? PFP_FLAG_SSA : 0) | (e ? PFP_FLAG_ESA : 0);
  double start0 = now_s(), start = start0;
  int status = 0;
  fprintf(logf, "==== %s\n==== input %s (%llu bytes) -w %d -p %llu%s%s%s\n", pfp_version(), input,
          (unsigned long long)n, w, p, s ? " -s" : "", e ? " -e" : "", S ? " -S" : "");
  if (gpus > 1 && getenv("PFP_MULTI_PYTHON") && atoi(getenv("PFP_MULTI_PYTHON"))) {
    /* the ranks come first: a process that has initialised the GPU must not start other programs on it */
    const char *textfile = input;
    char seqname[4096 + 16];
    if (fasta) {   /* the ranks read byte ranges of the filtered text */
      snprintf(seqname, sizeof seqname, "%s.seq", input);
      if (write_file(input, "seq", text, n)) return 1;
      textfile = seqname;
    }
    fflush(logf);
    printf("==== Parsing, BWT of parsing, final BWT on %d GPUs. Command: dist_main.py x %d (%s, -w %d -p %llu%s%s%s)\n", gpus,
           gpus, input, w, p, s ? " -s" : "", e ? " -e" : "", S ? " -S" : "");
    int code = run_ranks(gpus, textfile, input, w, p, flags, halo, verbose);
    if (fasta) unlink(seqname);
    if (code) {
      printf("Error executing command line:\n\t%d ranks of dist_main.py: exit code %d\nCheck log file: %s\n", gpus, code, logname);
      fprintf(logf, "error: %d ranks of dist_main.py ended with exit code %d\n", gpus, code);
      fclose(logf);
      return 1;
    }
    printf("Elapsed time: %.4f\n", now_s() - start);
  } else if (gpus > 1) {
    /* one thread and one context per GPU inside the library, RCCL between them */
    int devs[64];
    if (gpus > 64) { printf("-G: at most 64 GPUs\n"); return 2; }
    for (int g = 0; g < gpus; g++) devs[g] = device + g;
    pfp_multi_stats ms;
    char err[1024] = "";
    printf("==== Parsing, BWT of parsing, final BWT on %d GPUs. Command: pfp_bigbwt_files_multi(%s, -w %d -p %llu%s%s%s)\n", gpus,
           input, w, p, s ? " -s" : "", e ? " -e" : "", S ? " -S" : "");
    int code = pfp_bigbwt_files_multi(gpus, devs, text, n, w, p, flags, halo, input, &ms, err, sizeof err);
    if (code) {
      printf("Error executing command line:\n\t%s: %s\nCheck log file: %s\n", pfp_strerror(code), err, logname);
      fprintf(logf, "error %d: %s\n", code, err);
      fclose(logf);
      return 1;
    }
    fprintf(logf, "Ranks: %llu\nFound %llu distinct words\nTotal number of words: %llu\nDictionary size: %llu\nIndex width: %llu bits\n",
            (unsigned long long)ms.ranks, (unsigned long long)ms.n_words, (unsigned long long)ms.n_phrases,
            (unsigned long long)ms.dict_size, (unsigned long long)ms.index_bits);
    if (verbose)
      printf("  %llu ranks: chain %.1f ms, with files %.1f ms; suffix array of the dictionary in %llu share(s), of the parse in %llu\n",
             (unsigned long long)ms.ranks, ms.ms_chain, ms.ms_total, (unsigned long long)ms.sa_shares, (unsigned long long)ms.parse_shares);
    printf("Elapsed time: %.4f\n", now_s() - start);
  }

  pfp_ctx *ctx = NULL;
  const double t_ctx0 = now_s();
  int rc = pfp_ctx_create(&ctx, device);
  if (getenv("PFP_TRACE_HOST")) fprintf(stderr, "[pfp] driver: context after %.3f s (pfp_ctx_create %.3f s)\n", now_s() - t_main, now_s() - t_ctx0);
  if (rc) {
    fprintf(stderr, "Cannot initialise the GPU (%s): this tool has no CPU path\n", pfp_strerror(rc));
    return 1;
  }
  pfp_set_profiling(ctx, verbose);
  if (density > 0 && pfp_set_parse_density(ctx, density) != 0) { fprintf(stderr, "--density must lie in [0.01, 64]\n"); return 2; }

  if (gpus > 1) {
    /* done above, by the ranks */
  } else if (keep || parsing || compress) {
    /* staged run: materialise the reference's temp files (bigbwt -k / --parsing) */
    pfp_parse_result pr;
    printf("==== Parsing. Command: pfp_parse(%s, -w %d -p %llu%s)\n", input, w, p, flags ? " -s" : "");
    rc = pfp_parse(ctx, text, n, w, p, flags != 0, &pr);
    if (rc) goto fail;
    fprintf(logf, "Found %llu distinct words\nTotal number of words: %llu\n", (unsigned long long)pr.n_words,
            (unsigned long long)pr.n_phrases);
    status |= write_file(input, "parse", pr.parse, pr.n_phrases * 4);
    if (flags) status |= write_file(input, "sai", pr.sai, pr.n_phrases * 5);
    if (parsing || compress) {
      /* bigbwt:81 passes -c to the parser for both: the dictionary is written without the overlaps
       * (.dicz, what unparse reads); bigbwt:88-94 then removes .last/.occ, which are not written here */
      uint8_t *z = malloc(pr.dict_size + 1);
      if (!z) { fprintf(stderr, "out of memory\n"); return 1; }
      status |= write_file(input, "dicz", z, pfp_dicz_from_dict(pr.dict, pr.dict_size, w, z));
      free(z);
      printf("Elapsed time: %.4f\n", now_s() - start);
      pfp_parse_result_free(&pr);
      if (parsing) {
        printf("==== Stopping after the parsing phase as requested\n");
        goto done;
      }
      /* bigbwt:95-105 */
      start = now_s();
      char cmd[3 * 4096 + 64];
      snprintf(cmd, sizeof cmd, "XZ_OPT=-9 tar -cJf '%s.parse.txz' '%s.parse' '%s.dicz'", input, input, input);
      printf("==== Compressing. Command: %s\n", cmd);
      if (system(cmd) != 0) { printf("Error executing command line:\n\t%s\nCheck log file: %s\n", cmd, logname); status = 1; goto done; }
      printf("Elapsed time: %.4f\n", now_s() - start);
      if (!keep) {   /* delete_temp_files, bigbwt:201-217: .dicz stays */
        char nm[4096 + 16];
        printf("==== Deleting temporary files.\n");
        snprintf(nm, sizeof nm, "%s.parse", input); unlink(nm);
        if (s || S) { snprintf(nm, sizeof nm, "%s.sai", input); unlink(nm); }   /* bigbwt:210: not for -e alone */
      }
      printf("==== Done: Parsing output xz-compressed as requested\n");
      fclose(logf);
      pfp_ctx_destroy(ctx);
      return status ? 1 : 0;
    }
    status |= write_file(input, "dict", pr.dict, pr.dict_size);
    status |= write_file(input, "occ", pr.occ, pr.n_words * 4);
    status |= write_file(input, "last", pr.last, pr.n_phrases);
    printf("Elapsed time: %.4f\n", now_s() - start);
    start = now_s();
    uint64_t P = pr.n_phrases;
    uint32_t *ilist = malloc((P + 1) * 4);
    uint8_t *bwlast = malloc(P + 1), *bwsai = flags ? malloc((P + 1) * 5) : NULL;
    printf("==== Computing BWT of parsing. Command: pfp_bwtparse(%s%s)\n", input, flags ? " -s" : "");
    rc = pfp_bwtparse(ctx, pr.parse, P, pr.last, flags ? pr.sai : NULL, pr.occ, pr.n_words, ilist, bwlast, bwsai);
    if (rc) goto fail;
    status |= write_file(input, "ilist", ilist, (P + 1) * 4);
    status |= write_file(input, "bwlast", bwlast, P + 1);
    if (flags) status |= write_file(input, "bwsai", bwsai, (P + 1) * 5);
    printf("Elapsed time: %.4f\n", now_s() - start);
    start = now_s();
    pfp_bwt_result br;
    printf("==== Computing final BWT. Command: pfp_merge(-w %d %s%s%s%s)\n", w, input, s ? " -s" : "", e ? " -e" : "",
           S ? " -S" : "");
    rc = pfp_merge(ctx, pr.dict, pr.dict_size, pr.occ, pr.n_words, ilist, bwlast, bwsai, P + 1, w, flags, &br);
    if (rc) goto fail;
    status |= write_file(input, "bwt", br.bwt, br.bwt_size);
    if (S) status |= write_file(input, "sa", br.sa, br.sa_bytes);
    if (s) status |= write_file(input, "ssa", br.ssa, br.ssa_bytes);
    if (e) status |= write_file(input, "esa", br.esa, br.esa_bytes);
    printf("Elapsed time: %.4f\n", now_s() - start);
    pfp_bwt_result_free(&br);
    pfp_parse_result_free(&pr);
    free(ilist); free(bwlast); free(bwsai);
  } else {
    /* the text (an mmap of the input) is streamed to the GPU in chunks, the outputs are streamed from HBM straight
     * into input.bwt / .sa / .ssa / .esa: no output is held in host memory */
    printf("==== Parsing, BWT of parsing, final BWT on the GPU. Command: pfp_bigbwt_files(%s, -w %d -p %llu%s%s%s)\n", input, w,
           p, s ? " -s" : "", e ? " -e" : "", S ? " -S" : "");
    rc = fd_in >= 0 ? pfp_bigbwt_fd(ctx, fd_in, 0, n, w, p, flags, input, NULL) : pfp_bigbwt_files(ctx, text, n, w, p, flags, input, NULL);
    if (rc) goto fail;
    pfp_stats st;
    pfp_get_stats(ctx, &st);
    fprintf(logf, "Found %llu distinct words\nTotal number of words: %llu\nDictionary size: %llu\n"
                  "SA rounds dict/parse: %llu/%llu\nHard groups: %llu\nHard bwt chars: %llu\n",
            (unsigned long long)st.n_words, (unsigned long long)st.n_phrases, (unsigned long long)st.dict_size,
            (unsigned long long)st.sa_rounds_dict, (unsigned long long)st.sa_rounds_parse,
            (unsigned long long)st.hard_groups, (unsigned long long)st.hard_chars);
    if (verbose)
      printf("  GPU ms: scan %.2f phrases %.2f dictSA %.2f parseSA %.2f merge %.2f total %.2f\n", st.ms_scan,
             st.ms_phrases, st.ms_sa_dict, st.ms_sa_parse, st.ms_merge, st.ms_total);
    if (st.n != n) fprintf(stderr, "Invalid char found in input file: no additional chars will be read\n");
    fprintf(logf, "Index width: %llu bits\n", (unsigned long long)st.index_bits);      /* the reference's 32- / 64-bit executables (bigbwt:109-151) */
    printf("Elapsed time: %.4f\n", now_s() - start);
  }
  printf("Total construction time: %.4f\n", now_s() - start0);
  if (sum) {   /* bigbwt:160-171 */
    print_digest("BWT", input, "bwt");
    if (S) print_digest("SA ", input, "sa");
    if (s) print_digest("SSA", input, "ssa");
    if (e) print_digest("ESA", input, "esa");
  }
  if (!keep) printf("==== Deleting temporary files.\n");   /* nothing was written: intermediates lived in HBM */
  if (check) {   /* bigbwt:177-194: whole-text suffix array -> .Bwt, then compare */
    start = now_s();
    printf("==== Computing BWT using the whole-text suffix array. Command: pfp_sacak(%s)\n", input);
    /* simplebwt below 2^32 - 16 bytes, simplebwt64 (64-bit suffix array entries) above (bigbwt:177-182) */
    const int wide = n + 1 >= 0xFFFFFFF0ull;
    uint8_t *t0 = malloc(n + 1);
    uint32_t *SA = wide ? NULL : malloc((n + 1) * sizeof *SA);
    uint64_t *SA64 = wide ? malloc((n + 1) * sizeof *SA64) : NULL;
    uint8_t *B = malloc(n + 1);
    if (!t0 || (!SA && !SA64) || !B) { fprintf(stderr, "-c: out of memory\n"); status = 1; goto done; }
    memcpy(t0, text, n); t0[n] = 0;
    rc = wide ? pfp_sacak64(ctx, t0, SA64, n + 1) : pfp_sacak(ctx, t0, SA, n + 1);
    if (rc) goto fail;
    for (uint64_t i = 0; i <= n; i++) { const uint64_t v = wide ? SA64[i] : SA[i]; B[i] = v ? t0[v - 1] : 0; }   /* simplebwt.c:80-93 */
    free(SA64);
    status |= write_file(input, "Bwt", B, n + 1);
    printf("Elapsed time: %.4f\n", now_s() - start);
    char nm[4096];
    snprintf(nm, sizeof nm, "%s.bwt", input);
    FILE *f = fopen(nm, "rb");
    uint8_t *mine = malloc(n + 2);
    size_t got = f ? fread(mine, 1, n + 2, f) : 0;
    if (f) fclose(f);
    printf("==== Comparing BWTs. Command: cmp %s.bwt %s.Bwt\n", input, input);
    printf((got == n + 1 && memcmp(mine, B, n + 1) == 0) ? "BWTs match\n" : "BWTs differ\n");
    free(mine); free(t0); free(SA); free(B);
  }
done:
  printf("==== Done\n");
  fclose(logf);
  {
    const double t_d0 = now_s();
    pfp_ctx_destroy(ctx);
    if (getenv("PFP_TRACE_HOST")) fprintf(stderr, "[pfp] driver: pfp_ctx_destroy %.3f s, main() %.3f s\n", now_s() - t_d0, now_s() - t_main);
  }
  return status ? 1 : 0;
fail:
  /* bigbwt:235-239 */
  printf("Error executing command line:\n\t%s: %s\nCheck log file: %s\n", pfp_strerror(rc), pfp_last_error(ctx), logname);
  fprintf(logf, "error %d: %s\n", rc, pfp_last_error(ctx));
  fclose(logf);
  pfp_ctx_destroy(ctx);
  return 1;
}
