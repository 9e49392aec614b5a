#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
extern "C" {
void *icgh_batch_create(int device, int n_streams, const double *cam10, int width, int height, int max_features, double min_parallax, double max_interval,
                        int check_hist, double reproj_std, int window, int host_threads, int groups, char *err, int errlen);
int icgh_batch_run(void *b, int K, const void *const *images, int stride, int channels, int on_device, const double *stamps, const double *poses12,
                   int32_t *states, char *err, int errlen);
void icgh_batch_destroy(void *b);
int icgh_batch_stats(void *b, int stream, uint64_t *out8);
}
int main() {
    const int w = 640, h = 480, B = 6, ring = 12, K = 30;
    std::vector<uint8_t> fr((size_t) B * ring * w * h);
    std::vector<double> po((size_t) B * ring * 12), cam(10);
    FILE *f = fopen("/tmp/icg_tsan/fe_frames.bin", "rb"); if (fread(fr.data(), 1, fr.size(), f) != fr.size()) return 3; fclose(f);
    f = fopen("/tmp/icg_tsan/fe_poses.bin", "rb"); if (fread(po.data(), 8, po.size(), f) != po.size()) return 3; fclose(f);
    f = fopen("/tmp/icg_tsan/fe_cam.bin", "rb"); if (fread(cam.data(), 8, 10, f) != 10) return 3; fclose(f);
    char err[512] = {0};
    void *b = icgh_batch_create(0, B, cam.data(), w, h, 120, 20.0, 0.5, 0, 1.5, 10, 2, 3, err, 512);
    if (!b) { printf("create failed: %s\n", err); return 1; }
    std::vector<const void *> img((size_t) K * B);
    std::vector<double> st((size_t) K * B), ps((size_t) K * B * 12);
    std::vector<int32_t> states((size_t) K * B);
    for (int k = 0; k < K; k++) {
        int period = 2 * (ring - 1), m = k % period, fi = m < ring ? m : period - m;
        for (int s = 0; s < B; s++) {
            img[(size_t) k * B + s] = &fr[((size_t) s * ring + fi) * w * h];
            st[(size_t) k * B + s]  = 1000.0 + k / 20.0;
            for (int c = 0; c < 12; c++) ps[((size_t) k * B + s) * 12 + c] = po[((size_t) s * ring + fi) * 12 + c];
        }
    }
    // two calls: the second one re-uses the worker threads and frees frames created on the main thread on the workers
    for (int rep = 0; rep < 2; rep++) {
        int rc = icgh_batch_run(b, K / 2, img.data() + (size_t) rep * (K / 2) * B, w, 1, 0, st.data() + (size_t) rep * (K / 2) * B,
                                ps.data() + (size_t) rep * (K / 2) * B * 12, states.data(), err, 512);
        if (rc) { printf("run failed: %s\n", err); return 2; }
    }
    for (int s = 0; s < B; s++) { uint64_t o[8]; icgh_batch_stats(b, s, o); printf("stream %d: frames %lu keyframes %lu tracked %lu\n", s, o[0], o[1], o[2]); }
    icgh_batch_destroy(b);
    printf("done\n");
    return 0;
}
