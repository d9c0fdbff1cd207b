/* fasta.c -- see fasta.h.  A record starts at '>' or '@'; its header runs to the end of the line;
 * sequence lines follow until a line that starts with '>', '@' or '+'.  After '+' (FASTQ) the rest
 * of that line and as many quality characters as there were bases are skipped, then the reader
 * hunts for the next '>' or '@' byte.  Inside sequence lines '\n' is dropped and a '\r' that ends
 * a line is dropped (kseq.h:141). */
#include "fasta.h"
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

size_t pfp_fasta_text(const uint8_t *in, size_t n, uint8_t *out) {
  size_t pos = 0, o = 0;
  int last_char = 0;
  for (;;) {
    if (last_char == 0) {                       /* hunt for the next header byte */
      int c = -1;
      while (pos < n) { c = in[pos++]; if (c == '>' || c == '@') break; c = -1; }
      if (c < 0) return o;
      last_char = c;
    }
    /* header: name up to white space, then the rest of the line */
    if (pos >= n) return o;
    while (pos < n && !isspace(in[pos])) pos++;
    if (pos < n) { int d = in[pos++]; if (d != '\n') { while (pos < n && in[pos] != '\n') pos++; if (pos < n) pos++; } }
    /* sequence lines */
    size_t rec0 = o;
    int c = -1;
    while (pos < n) {
      c = in[pos++];
      if (c == '>' || c == '+' || c == '@') break;
      if (c == '\n') { c = -1; continue; }
      out[o++] = (uint8_t)c;
      c = -1;
      if (pos >= n) break;                       /* nothing after it: no line to finish (kseq.h:102-106) */
      while (pos < n && in[pos] != '\n') out[o++] = in[pos++];
      if (pos < n) pos++;
      if (o - rec0 > 1 && out[o - 1] == '\r') o--;
    }
    if (c == '>' || c == '@') last_char = c;
    size_t seqlen = o - rec0;
    if (c == '+') {                              /* FASTQ: skip the '+' line and the quality string */
      int nl = 0;
      while (pos < n) { if (in[pos++] == '\n') { nl = 1; break; } }
      if (!nl) { o = rec0; return o; }           /* kseq returns -2: the record is not delivered */
      size_t ql = 0;
      while (pos < n) {                          /* at least one line is read, kseq.h:218 */
        size_t s = pos;
        while (pos < n && in[pos] != '\n') pos++;
        size_t e = pos;
        if (pos < n) pos++;
        ql += e - s;
        if (ql > 1 && e > s && in[e - 1] == '\r') ql--;
        if (ql >= seqlen) break;
      }
      last_char = 0;
      if (ql != seqlen) { o = rec0; return o; }
    }
    /* newscan.cpp:338-349: upper-case, stop at the first byte <= Dollar */
    for (size_t i = rec0; i < o; i++) {
      int u = toupper(out[i]);
      if (u <= 2) return i;
      out[i] = (uint8_t)u;
    }
    if (pos >= n && c != '>' && c != '@') return o;
  }
}

uint8_t *pfp_read_maybe_gz(const char *path, size_t *n) {
  gzFile f = gzopen(path, "rb");
  if (!f) return NULL;
  size_t cap = 1 << 24, len = 0;
  uint8_t *buf = malloc(cap);
  for (;;) {
    if (!buf) { gzclose(f); return NULL; }
    int r = gzread(f, buf + len, (unsigned)(cap - len > (1u << 30) ? (1u << 30) : cap - len));
    if (r < 0) { free(buf); gzclose(f); return NULL; }
    if (r == 0) break;
    len += (size_t)r;
    if (len == cap) { cap *= 2; buf = realloc(buf, cap); }
  }
  gzclose(f);
  *n = len;
  return buf;
}

size_t pfp_dicz_from_dict(const uint8_t *dict, size_t dict_size, int w, uint8_t *out) {
  size_t o = 0, s = 0;
  for (size_t i = 0; i + 1 < dict_size; i++) {
    if (dict[i] != 1) continue;
    size_t b = s, e = i - (size_t)w;
    if (dict[b] == 2) b++;
    memcpy(out + o, dict + b, e - b);
    o += e - b;
    out[o++] = 1;
    s = i + 1;
  }
  out[o++] = 0;
  return o;
}
