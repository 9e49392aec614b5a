#include "../../ic-gvins_amd/host/object_pool.h"
#include <thread>
#include <vector>
#include <mutex>
#include <cstdio>
#include <cstdint>
using namespace icg;
struct S { uint64_t a, b; };
int main() {
    const int T = 4, N = 60000;
    std::vector<std::mutex> m(T);
    std::vector<std::vector<S*>> box(T);
    auto w = [&](int t) {
        std::vector<S*> mine;
        for (int i = 0; i < N; i++) {
            S *s = PoolAllocator<S>().allocate(1); s->a = i; s->b = ~(uint64_t) i;
            if (i % 3 == 0) { std::lock_guard<std::mutex> l(m[(t + 1) % T]); box[(t + 1) % T].push_back(s); } else mine.push_back(s);
            if (i % 128 == 127) {
                std::vector<S*> got; { std::lock_guard<std::mutex> l(m[t]); got.swap(box[t]); }
                for (S *g : got) { if (g->b != ~g->a) printf("BAD\n"); PoolAllocator<S>().deallocate(g, 1); }
                for (S *g : mine) PoolAllocator<S>().deallocate(g, 1);
                mine.clear();
            }
        }
        for (S *g : mine) PoolAllocator<S>().deallocate(g, 1);
    };
    std::vector<std::thread> th; for (int t = 0; t < T; t++) th.emplace_back(w, t); for (auto &x : th) x.join();
    printf("done\n");
}
