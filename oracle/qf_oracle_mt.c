/*
 * qf_oracle_mt.c -- OpenMP "all host cores" variants of the CPU restatement, used only by
 * bench.py's cpu_baseline leg (TEST INFRASTRUCTURE, see qf_oracle.c).  The reference is
 * single-threaded; these run the same one-message-per-call functions over independent
 * messages on every core, which is the most a user of the reference could do by hand.
 */
#include <stdint.h>
#include <stddef.h>
#include <omp.h>

void qfo_matmul_batch(const uint32_t *A, uint32_t rows, uint32_t cols, uint32_t q, const void *X,
                      int xkind, size_t xstride_bytes, uint64_t B, const uint32_t *addend,
                      const uint8_t *msg, uint32_t *out);
void qfo_dot_decode_batch(const uint32_t *s, uint32_t n, uint32_t q, const uint32_t *C,
                          uint64_t B, uint8_t *bits);
int qfo_polymul_batch(const uint32_t *a, const uint32_t *b, uint32_t n, uint32_t q, uint64_t B,
                      int use_ntt, uint32_t *c);

int qfo_num_threads(void) { return omp_get_max_threads(); }

void qfo_matmul_batch_mt(const uint32_t *A, uint32_t rows, uint32_t cols, uint32_t q,
                         const void *X, int xkind, size_t xstride_bytes, uint64_t B,
                         const uint32_t *addend, const uint8_t *msg, uint32_t *out) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++)
        qfo_matmul_batch(A, rows, cols, q, (const uint8_t *)X + b * xstride_bytes, xkind,
                         xstride_bytes, 1, addend ? addend + b * rows : 0, msg ? msg + b : 0,
                         out + b * rows);
}
void qfo_dot_decode_batch_mt(const uint32_t *s, uint32_t n, uint32_t q, const uint32_t *C,
                             uint64_t B, uint8_t *bits) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < (int64_t)B; b++)
        qfo_dot_decode_batch(s, n, q, C + b * (n + 1), 1, bits + b);
}
int qfo_polymul_batch_mt(const uint32_t *a, const uint32_t *b, uint32_t n, uint32_t q, uint64_t B,
                         int use_ntt, uint32_t *c) {
    int rc = 0;
    int T = omp_get_max_threads();
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; t++) {
        uint64_t lo = B * t / T, hi = B * (t + 1) / T;
        if (hi > lo && qfo_polymul_batch(a + lo * n, b + lo * n, n, q, hi - lo, use_ntt, c + lo * n))
            rc = -1;
    }
    return rc;
}
