// Dev probe: which physical CUs (XCC, SE, CU) does a stream created with hipExtStreamCreateWithCUMask run on, for a few mask patterns?
// build: hipcc --offload-arch=gfx950 -O2 -o tools/dev/_build/cumask_probe tools/dev/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
__global__ void probe(unsigned *out, int spin) {
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        out[2 * blockIdx.x] = hw, out[2 * blockIdx.x + 1] = xcc;
    }
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
static void run(const char *name, std::vector<uint32_t> mask, int nwg) {
    hipStream_t s;
    hipError_t e = mask.empty() ? hipStreamCreateWithFlags(&s, hipStreamNonBlocking) : hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    unsigned *d;
    hipMalloc(&d, nwg * 8);
    hipMemsetAsync(d, 0, nwg * 8, s);
    probe<<<nwg, 64, 0, s>>>(d, 20000);
    std::vector<unsigned> h(nwg * 2);
    hipMemcpyAsync(h.data(), d, nwg * 8, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    std::map<int, std::set<int>> per;
    for (int i = 0; i < nwg; ++i) {
        unsigned hw = h[2 * i], x = h[2 * i + 1] & 15;
        int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[x].insert(se * 32 + sh * 16 + cu);
    }
    int tot = 0;
    printf("%s:", name);
    for (auto &kv : per) {
        printf(" xcc%d=%zu[", kv.first, kv.second.size());
        for (int v : kv.second) printf("%d.%d ", v / 32, v % 16);
        printf("]");
        tot += (int)kv.second.size();
    }
    printf(" total=%d\n", tot);
    hipFree(d);
    hipStreamDestroy(s);
}
int main() {
    const int nwg = 8192;
    run("nomask", {}, nwg);
    std::vector<uint32_t> m(8, 0);
    auto pat = [&](auto f) { std::vector<uint32_t> v(8, 0); for (int i = 0; i < 256; ++i) if (f(i)) v[i / 32] |= 1u << (i % 32); return v; };
    run("first80", pat([](int i) { return i < 80; }), nwg);
    run("first8", pat([](int i) { return i < 8; }), nwg);
    run("bits8..15", pat([](int i) { return i >= 8 && i < 16; }), nwg);
    run("mod32<10", pat([](int i) { return i % 32 < 10; }), nwg);
    run("mod8==0", pat([](int i) { return i % 8 == 0; }), nwg);
    run("last176", pat([](int i) { return i >= 80; }), nwg);
    return 0;
}
