/* selftest.c -- runs the oracle against the reference's known-answer vectors
 * natively, so that it can be built with -fsanitize=address,undefined
 * (tests/test_oracle_sanitize.py).  Test infrastructure only. */
#include "hark_oracle.c"
#include <stdio.h>

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); fails++; } } while (0)

static int eq_i32(const int32_t *a, const int32_t *b, int64_t n) { return n == 0 || memcmp(a, b, (size_t)n * 4) == 0; }

int main(void)
{
    {   /* segmented_tests.fut:7-8, :19-20 */
        uint8_t f[10] = {1,0,0,1,0,0,1,0,0,0}; int32_t a[10] = {1,2,3,4,5,6,7,8,9,10}, out[10]; int64_t n = 0;
        int32_t exp[10] = {1,3,6,4,9,15,7,15,24,34}, expr[3] = {6,15,34};
        CHECK(ora_segmented_scan_add_i32(f, a, 10, out) == 0 && eq_i32(out, exp, 10));
        CHECK(ora_segmented_reduce_add_i32(f, a, 10, out, &n) == 0 && n == 3 && eq_i32(out, expr, 3));
        CHECK(ora_segmented_scan_add_i32(f, a, 0, out) == 0);                      /* :11-12 empty */
        CHECK(ora_segmented_reduce_add_i32(f, a, 1, out, &n) == 0 && n == 1 && out[0] == 1);   /* :21-22 */
    }
    {   /* :29-34 replicated_iota */
        int32_t r1[3] = {2,3,1}, e1[6] = {0,0,1,1,1,2}, r3[3] = {2,0,1}, e3[3] = {0,0,2}, z2[2] = {0,0}, out[8]; int64_t n;
        CHECK(ora_replicated_iota(r1, 3, out, &n) == 0 && n == 6 && eq_i32(out, e1, 6));
        CHECK(ora_replicated_iota(r3, 3, out, &n) == 0 && n == 3 && eq_i32(out, e3, 3));
        CHECK(ora_replicated_iota(z2, 2, out, &n) == 0 && n == 0);
        CHECK(ora_replicated_iota(z2, 0, out, &n) == 0 && n == 0);
    }
    {   /* :41-45 segmented_iota, :52-53 expand, :60-61, :68-69 */
        uint8_t f[7] = {0,0,0,1,0,0,0}; int32_t e[7] = {0,1,2,0,1,2,3}, out[8]; int64_t n;
        CHECK(ora_segmented_iota(f, 7, out) == 0 && eq_i32(out, e, 7));
        int32_t a[3] = {2,3,1}, ee[6] = {0,2,0,3,6,0};
        CHECK(ora_test_expand(a, 3, out, &n) == 0 && n == 6 && eq_i32(out, ee, 6));
        int32_t b[4] = {2,0,3,1}, er[3] = {2,9,0}, eo[4] = {2,0,9,0};
        CHECK(ora_test_expand_reduce(b, 4, out, &n) == 0 && n == 3 && eq_i32(out, er, 3));
        CHECK(ora_test_expand_outer_reduce(b, 4, out, &n) == 0 && n == 4 && eq_i32(out, eo, 4));
    }
    {   /* operator goldens on data.csv (G1, G2, G4, G9 count) */
        uint32_t db[7][8] = {{6,6,6,6,6,6,6,6},{0,0,0,0,0,0,0,0},{0,0,0,0,0,0,0,0},{0,0,0,0,0,0,0,0},{0,0,0,0,0,0,0,0},{6,6,6,6,6,6,6,6},{1,2,3,4,5,3,2,1}};
        int32_t cols[2] = {0, 2}, sel[14];
        CHECK(ora_query_sel((const int32_t *)db, 7, 8, cols, 2, sel) == 0 && sel[12] == 1 && sel[13] == 3);
        int32_t t_cols[2] = {0, 3}; uint32_t *res = NULL; int64_t g = 0;
        CHECK(ora_query_groupby(&db[0][0], 7, 8, 0, cols, 2, t_cols, 2, &res, &g) == 0 && g == 3);
        uint32_t g2[9] = {0,0,0, 1,1,3, 6,6,6};
        CHECK(res && memcmp(res, g2, sizeof g2) == 0);
        ora_free(res);
        int32_t s4[3] = {6,6,6}, t4[3] = {2,1,4}; uint32_t g4[12] = {0,0,0,0, 1,2,2,2, 6,12,36,6};
        CHECK(ora_query_groupby(&db[0][0], 7, 8, 0, s4, 3, t4, 3, &res, &g) == 0 && g == 3 && memcmp(res, g4, sizeof g4) == 0);
        ora_free(res);
        int32_t c1[2] = {0, 2}, c2[1] = {7}; int64_t p = 0;
        CHECK(ora_join(&db[0][0], 7, 8, &db[0][0], 7, 8, 0, 0, c1, 2, c2, 1, &res, &p) == 0 && p == 21);
        CHECK(res && res[16 * 3] == 1 && res[16 * 3 + 1] == 3 && res[16 * 3 + 2] == 1);
        ora_free(res);
        CHECK(ora_query_groupby(&db[0][0], 0, 8, 0, cols, 2, t_cols, 2, &res, &g) == 0 && g == 0 && res == NULL);   /* G6 */
        int32_t badc[1] = {8};
        CHECK(ora_query_sel((const int32_t *)db, 7, 8, badc, 1, sel) == ORA_EBOUNDS);
    }
    {   /* extension oracles: the reference-algorithm port agrees with the direct fold */
        enum { N = 5000, G = 37 };
        static float p[N], v[N], s32[G]; static int32_t k[N]; static double s64[G]; static int64_t cnt[G];
        ora_gen_columns(42, 0, N, G, 1, p, k, v);
        CHECK(ora_filter_groupby_dense_f32(p, k, v, N, CMP_GT, 0.5f, G, s32, s64, cnt) == 0);
        uint32_t *kk, *cc; float *ss; int64_t g;
        CHECK(ora_filter_groupby_refalgo_f32(p, k, v, N, CMP_GT, 0.5f, &kk, &ss, &cc, &g) == 0 && g == G);
        for (int64_t i = 0; i < g; i++) CHECK(kk[i] == (uint32_t)i && ss[i] == s32[i] && (int64_t)cc[i] == cnt[i]);
        ora_free(kk); ora_free(ss); ora_free(cc);
        static int64_t idx[N];
        CHECK(ora_filter_f32(p, N, CMP_LE, 0.5f, idx) + (int64_t)0 >= 0);
    }
    printf(fails ? "oracle selftest: %d FAILED\n" : "oracle selftest: ok\n", fails);
    return fails ? 1 : 0;
}
