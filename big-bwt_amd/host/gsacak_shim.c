/* gsacak_shim.c -- libpfpgsacak.so / libpfpgsacak64.so: the reference's sacak / sacak_int / gsacak symbols
 * (gsa/gsacak.h:78-105) over libpfpgpu.so, see include/pfpgsacak.h.  Compiled twice: plain and with -DM64. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include "pfpgsacak.h"
#include "pfpgpu.h"

static pfp_ctx *g_ctx;
static int g_rc;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static void shim_destroy(void) {
  if (g_ctx) { pfp_ctx_destroy(g_ctx); g_ctx = NULL; }
}
static void shim_create(void) {
  const char *e = getenv("PFP_GSACAK_DEVICE");
  g_rc = pfp_ctx_create(&g_ctx, e ? atoi(e) : 0);
  if (g_rc != PFP_OK) {
    fprintf(stderr, "libpfpgsacak: no GPU context (%s); there is no CPU fallback\n", pfp_strerror(g_rc));
    g_ctx = NULL;
    return;
  }
  atexit(shim_destroy);
}
/* the process-wide context, locked; NULL (and nothing locked) when it could not be made */
static pfp_ctx *shim_enter(void) {
  pthread_once(&g_once, shim_create);
  if (!g_ctx) return NULL;
  pthread_mutex_lock(&g_mu);
  return g_ctx;
}
static int shim_leave(pfp_ctx *c, int rc, const char *what) {
  if (rc != PFP_OK) fprintf(stderr, "libpfpgsacak: %s failed: %s (%s)\n", what, pfp_strerror(rc), pfp_last_error(c));
  pthread_mutex_unlock(&g_mu);
  return rc == PFP_OK ? 0 : -1;
}

int sacak(unsigned char *s, uint_t *SA, uint_t n) {
  if (!s || !SA || n == 0) return -1;                  /* gsacak.c:2493 */
  pfp_ctx *c = shim_enter();
  if (!c) return -1;
#if M64
  return shim_leave(c, pfp_sacak64(c, s, SA, n), "sacak");
#else
  return shim_leave(c, pfp_sacak(c, s, SA, n), "sacak");
#endif
}

int sacak_int(int_text *s, uint_t *SA, uint_t n, uint_t k) {
  if (!s || !SA || n == 0) return -1;                  /* gsacak.c:2498 */
  pfp_ctx *c = shim_enter();
  if (!c) return -1;
#if M64
  return shim_leave(c, pfp_sacak_int64(c, s, SA, n, k), "sacak_int");
#else
  return shim_leave(c, pfp_sacak_int(c, s, SA, n, k), "sacak_int");
#endif
}

int gsacak(unsigned char *s, uint_t *SA, int_t *LCP, int_t *DA, uint_t n) {
  if (!s || !SA || n == 0) return -1;                  /* gsacak.c:2503 */
  pfp_ctx *c = shim_enter();
  if (!c) return -1;
#if M64
  return shim_leave(c, pfp_gsacak_lcp_da64(c, s, SA, LCP, DA, n), "gsacak");
#else
  return shim_leave(c, pfp_gsacak_lcp_da(c, s, SA, LCP, DA, n), "gsacak");
#endif
}
