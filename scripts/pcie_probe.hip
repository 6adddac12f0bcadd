// pinned host <-> device copy rates of this box (what the staging of hp_wfa2.hip can hope for); hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
int main(int argc, char** argv) {
    const bool only_h2d = argc > 1 && !strcmp(argv[1], "h2d");   // bench.py: just the pinned host -> device rate
    const size_t n = 1ull << 30;
    void *h = nullptr, *d = nullptr;
    hipHostMalloc(&h, n, hipHostMallocDefault); hipMalloc(&d, n);
    memset(h, 1, n);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("H2D 1 GiB pinned: %.1f GB/s\n", n / dt / 1e9);
    }
    if (only_h2d) return 0;
    for (size_t piece : {1ull << 22, 1ull << 24, 48ull << 20}) {
        auto t0 = std::chrono::steady_clock::now();
        for (size_t o = 0; o + piece <= n; o += piece) hipMemcpyAsync((char*)d + o, (char*)h + o, piece, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("H2D in %zu MiB pieces: %.1f GB/s\n", piece >> 20, n / dt / 1e9);
    }
    {
        auto t0 = std::chrono::steady_clock::now();
        hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("D2H 1 GiB pinned: %.1f GB/s\n", n / dt / 1e9);
    }
    std::vector<char> src(n, 2);
    for (int nt : {1, 4, 8, 16, 32}) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { memcpy((char*)h + n / nt * t, src.data() + n / nt * t, n / nt); });
        for (auto& x : th) x.join();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("host memcpy pageable -> pinned, %d threads: %.1f GB/s\n", nt, n / dt / 1e9);
    }
    return 0;
}
