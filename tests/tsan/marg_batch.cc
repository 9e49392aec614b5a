// Sanitizer driver 7 (tests/tsan/run.sh): MarginalizationBatch (host/marg_batch.h) — 12 windows, one of them on the dense path, marginalized three
// times on one batch object with the per-window phases on 4 pool threads — through the C entry point the tests use
// (icgh_backend_marginalize_batch, capi.cc), on the CPU backend of the C ABI.  Checks the result against the one-by-one mode.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" int icgh_backend_marginalize_batch(int mode, int n_windows, int dense_window, double jitter, int reps, int n, const double *obs_soa,
                                              const int32_t *idx_i, const int32_t *idx_j, const int32_t *idx_lm, int n_poses, const double *poses,
                                              const double *ext, int n_lm, const double *invdepth, double td, double huber_delta, double prior_weight,
                                              int host_threads, int32_t *sizes, double *Hp, double *bp, double *J0, double *e0, int32_t *counts,
                                              double *seconds, char *err, int errlen);

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[4];
    if (fread(hdr, sizeof hdr, 1, f) != 1) return 2;
    const int n = hdr[0], K = hdr[1], L = hdr[2];
    std::vector<double> obs((size_t) 15 * n), poses((size_t) 7 * K), ext(7), inv((size_t) L), td(1);
    std::vector<int32_t> ii((size_t) n), jj((size_t) n), ll((size_t) n);
    bool ok = fread(obs.data(), 8, obs.size(), f) == obs.size() && fread(ii.data(), 4, ii.size(), f) == ii.size() &&
              fread(jj.data(), 4, jj.size(), f) == jj.size() && fread(ll.data(), 4, ll.size(), f) == ll.size() &&
              fread(poses.data(), 8, poses.size(), f) == poses.size() && fread(ext.data(), 8, 7, f) == 7 &&
              fread(inv.data(), 8, inv.size(), f) == inv.size() && fread(td.data(), 8, 1, f) == 1;
    fclose(f);
    if (!ok) return 2;
    const int W = 12, cap = 6 * K + L + 7;
    std::vector<double> out[2][4];
    int32_t sizes[2][2], counts[2][2];
    for (int mode = 0; mode < 2; mode++) {
        for (int k = 0; k < 4; k++) out[mode][k].assign((size_t) W * cap * ((k & 1) ? 1 : cap), 0.0);
        double seconds = 0;
        char err[512] = {0};
        const int rc = icgh_backend_marginalize_batch(mode, W, 2, 1e-3, 3, n, obs.data(), ii.data(), jj.data(), ll.data(), K, poses.data(), ext.data(), L,
                                                      inv.data(), td[0], 1.0, 100.0, 4, sizes[mode], out[mode][0].data(), out[mode][1].data(),
                                                      out[mode][2].data(), out[mode][3].data(), counts[mode], &seconds, err, 512);
        if (rc != 0) {
            printf("mode %d failed: %d %s\n", mode, rc, err);
            return 1;
        }
    }
    const size_t r = (size_t) sizes[0][1];
    double worst = 0, scale = 0;
    for (size_t k = 0; k < (size_t) W * r * r; k++) scale = std::fmax(scale, std::fabs(out[1][0][k])), worst = std::fmax(worst, std::fabs(out[0][0][k] - out[1][0][k]));
    printf("%d windows, remained size %zu, structured/dense %d/%d (one by one %d/%d), max |Hp batch - Hp one by one| / max |Hp| = %.3g\n", W, r,
           counts[0][0], counts[0][1], counts[1][0], counts[1][1], worst / scale);
    if (!(worst <= 1e-9 * scale) || counts[0][0] != W - 1 || counts[0][1] != 1) return 1;
    printf("done\n");
    return 0;
}
