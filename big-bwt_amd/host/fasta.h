/* fasta.h -- the text `bigbwt -f` feeds to the parser (reference newscan.cpp:332-352 over
 * kseq.h:178-222): FASTA/FASTQ records, header lines and newlines dropped, sequences upper-cased
 * and concatenated with no separator; reading stops at the first byte <= 2. */
#ifndef PFP_FASTA_H
#define PFP_FASTA_H
#include <stddef.h>
#include <stdint.h>
/* out must hold n bytes; returns the number of bytes written */
size_t pfp_fasta_text(const uint8_t *in, size_t n, uint8_t *out);
/* reads a plain or gzip-compressed file completely (zlib gzread, as the reference does with
 * gzopen); returns a malloc'ed buffer or NULL */
uint8_t *pfp_read_maybe_gz(const char *path, size_t *n);
/* newscan.cpp:409-412 (-c): the dictionary without the overlaps -- every word loses its last w
 * chars and the first word its leading 0x02; out must hold dict_size bytes; returns bytes written */
size_t pfp_dicz_from_dict(const uint8_t *dict, size_t dict_size, int w, uint8_t *out);
#endif
