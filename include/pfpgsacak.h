/* pfpgsacak.h -- link-level replacement of the reference's in-process suffix sorter, gsa/gsacak.h:78-105.
 *
 * libpfpgsacak.so (32-bit build, uint_t = uint32_t) and libpfpgsacak64.so (the reference's -DM64 build,
 * gsa/gsacak.h:42-60: uint_t = uint64_t, int_t = int64_t, int_text stays 32 bits) export the reference's own three
 * symbols with the reference's own signatures, so that its callers
 *     bwtparse.c:167   sacak_int(Text, SA, n, k)
 *     pfbwt.cpp:495    gsacak(d, sa, lcp, NULL, dsize)
 *     simplebwt.c:77   sacak(Text, SA, n)
 * link against the GPU sorter WITHOUT a source edit:  gcc bwtparse.c utils.c -L big-bwt_amd -lpfpgsacak ...
 * (oracle/Makefile target `shim` does exactly that with the sources where they lie under /root/reference; the -m gpu
 * test tests/test_gsacak_shim.py runs those executables and compares their files with the reference-made goldens).
 *
 * Behaviour kept from gsa/gsacak.c:
 *   - return value >= 0 on success (the reference returns its recursion depth; callers only print it or test < 0),
 *     -1 when s or SA is NULL or n is 0 (gsacak.c:2493, 2498, 2503) - and -1, with a message on stderr, when the GPU
 *     library reports an error (no device, out of memory): nothing calls exit();
 *   - the caller owns and pre-allocates every array; LCP and DA are optional (NULL);
 *   - s[n-1] must be 0 (sacak / sacak_int: unique smallest; gsacak: separators are bytes 1, ordered by position);
 *   - re-entrant per call; calls from several threads are serialised on one process-wide context that is created
 *     on first use (device PFP_GSACAK_DEVICE, default 0) and lives until the process ends.
 * gsacak_int (gsacak.h:105) has no caller on the parse -> SA -> BWT path and is not provided.
 */
#ifndef PFPGSACAK_H
#define PFPGSACAK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef M64
#define M64 0
#endif
#if M64
typedef int64_t int_t;
typedef uint64_t uint_t;
#else
typedef int32_t int_t;
typedef uint32_t uint_t;
#endif
typedef uint32_t int_text;

int sacak(unsigned char *s, uint_t *SA, uint_t n);                               /* gsa/gsacak.h:86 */
int sacak_int(int_text *s, uint_t *SA, uint_t n, uint_t k);                      /* gsa/gsacak.h:92 */
int gsacak(unsigned char *s, uint_t *SA, int_t *LCP, int_t *DA, uint_t n);       /* gsa/gsacak.h:107 */

#ifdef __cplusplus
}
#endif
#endif
