/* bench_loop.c -- bench.py's step loop in native code: K back-to-back calls of asif_hip_filter_batch (or of
 * asif_hip_qp_solve_batch) with arguments marshalled once.  The library's callers are C++ programs; a Python / ctypes
 * call costs ~1.5 us of argument conversion per step, which is most of the explicit filter's 2 us kernel and would be
 * measured as if it were the library's.  Harness only: nothing here computes; built into libbench_loop.so. */
#include <stdint.h>

typedef int (*filter_fn)(void *, int64_t, int64_t, const double *, const double *, double *, double *, int32_t *,
                         double *, void *);

int bench_loop_filter(void *fn, int32_t k, void *ctx, int64_t B, int64_t ld, const double *x, const double *udes,
                      double *uact, double *relax, int32_t *rc, double *diag, void *stream)
{
	filter_fn f = (filter_fn)fn;
	for (int32_t i = 0; i < k; i++) {
		const int r = f(ctx, B, ld, x, udes, uact, relax, rc, diag, stream);
		if (r) return r;
	}
	return 0;
}

typedef int (*qp_fn)(int, const void *, int64_t, int64_t, int32_t, int32_t, const double *, const double *,
                     const double *, const double *, const double *, const double *, const uint8_t *, double *,
                     int32_t *, int32_t *, void *);

int bench_loop_qp(void *fn, int32_t k, int device, const void *solver, int64_t B, int64_t ld, int32_t nv, int32_t nc,
                  const double *Hd, const double *c, const double *A, const double *b, const double *lb,
                  const double *ub, const uint8_t *be, double *sol, int32_t *status, int32_t *iters, void *stream)
{
	qp_fn f = (qp_fn)fn;
	for (int32_t i = 0; i < k; i++) {
		const int r = f(device, solver, B, ld, nv, nc, Hd, c, A, b, lb, ub, be, sol, status, iters, stream);
		if (r) return r;
	}
	return 0;
}
