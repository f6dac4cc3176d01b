// What does a dependent step cost on this chip: a kernel boundary, or a grid-wide barrier inside one persistent kernel?
// (a) N dependent launches of a small kernel on one stream (each reads what the previous one wrote) - the kernel is empty enough
//     that this measures the host's enqueue rate - and (a') the same chain as one captured graph: the device-side gap;
// (b) ONE launch of G workgroups that runs the same N steps separated by a grid barrier (monotonic device-scope counter,
//     release/acquire fences; bounded spin: a barrier that does not complete within ~50 ms sets an abort flag and everyone leaves).
// Each step: every workgroup reads 4 KB written by ANOTHER workgroup in the previous step and writes its own 4 KB.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/grid_barrier_probe.hip -o tools/bin/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void step_body(const float* __restrict__ in, float* __restrict__ out, int wg, int nwg, int tid) {
    const int src = (wg * 7 + 3) % nwg;                    // someone else's block
    const float4 v = reinterpret_cast<const float4*>(in + (size_t)src * 1024)[tid];
    float4 o = {v.x + 1.f, v.y + 1.f, v.z + 1.f, v.w + 1.f};
    reinterpret_cast<float4*>(out + (size_t)wg * 1024)[tid] = o;
}

__global__ __launch_bounds__(256) void step_kernel(const float* in, float* out, int nwg) { step_body(in, out, blockIdx.x, nwg, threadIdx.x); }

__global__ __launch_bounds__(256) void persistent_kernel(float* a, float* b, int nsteps, unsigned* counter, int* abort_flag) {
    const int wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
    float* in = a; float* out = b;
    for (int s = 0; s < nsteps; ++s) {
        step_body(in, out, wg, nwg, tid);
        // grid barrier s: everyone's stores visible device-wide, then count, then wait for nwg * (s + 1)
        __syncthreads();
        if (tid == 0) {
            __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);          // agent scope: writes back this XCD's dirty L2 lines first
            const unsigned target = (unsigned)nwg * (unsigned)(s + 1);
            long spins = 0;
            while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 20000000l || __atomic_load_n(abort_flag, __ATOMIC_RELAXED)) { __atomic_store_n(abort_flag, 1, __ATOMIC_RELAXED); break; }
            }
        }
        __syncthreads();
        if (__atomic_load_n(abort_flag, __ATOMIC_RELAXED)) return;
        float* t = in; in = out; out = t;
    }
}

int main(int argc, char** argv) {
    const int nsteps = argc > 1 ? atoi(argv[1]) : 40;
    float *a, *b; unsigned* counter; int* abort_flag;
    CK(hipMalloc(&a, 1024 * 1024 * 4)); CK(hipMalloc(&b, 1024 * 1024 * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&abort_flag, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {8, 32, 64, 128, 256}) {
        CK(hipMemset(a, 0, 1024 * 1024 * 4)); CK(hipMemset(b, 0, 1024 * 1024 * 4));
        // (a) dependent launches
        float ms_l = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(a, 0, 1024 * 1024 * 4)); CK(hipMemset(b, 0, 1024 * 1024 * 4));
            CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0));
            for (int s = 0; s < nsteps; ++s) hipLaunchKernelGGL(step_kernel, dim3(G), dim3(256), 0, 0, (s & 1) ? b : a, (s & 1) ? a : b, G);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_l, e0, e1));
        }
        std::vector<float> ref(G * 1024); CK(hipMemcpy(ref.data(), (nsteps & 1) ? b : a, ref.size() * 4, hipMemcpyDeviceToHost));
        // (a') the same chain as one captured graph
        hipStream_t st; CK(hipStreamCreate(&st)); hipGraph_t g; hipGraphExec_t ge;
        CK(hipMemset(a, 0, 1024 * 1024 * 4)); CK(hipMemset(b, 0, 1024 * 1024 * 4)); CK(hipDeviceSynchronize());
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int s = 0; s < nsteps; ++s) hipLaunchKernelGGL(step_kernel, dim3(G), dim3(256), 0, st, (s & 1) ? b : a, (s & 1) ? a : b, G);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms_g = 0;
        for (int rep = 0; rep < 3; ++rep) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_g, e0, e1)); }
        // (b) one persistent launch
        float ms_p = 0; int bad = 0, aborted = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(a, 0, 1024 * 1024 * 4)); CK(hipMemset(b, 0, 1024 * 1024 * 4)); CK(hipMemset(counter, 0, 4)); CK(hipMemset(abort_flag, 0, 4));
            CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(256), 0, 0, a, b, nsteps, counter, abort_flag);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_p, e0, e1));
            CK(hipMemcpy(&aborted, abort_flag, 4, hipMemcpyDeviceToHost));
        }
        std::vector<float> got(G * 1024); CK(hipMemcpy(got.data(), (nsteps & 1) ? b : a, got.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
        printf("G=%3d workgroups, %d steps: launches %.2f us/step, graph %.2f us/step, persistent + grid barrier %.2f us/step  (%s, %d wrong values, expect %.0f got %.0f)\n",
               G, nsteps, ms_l * 1e3 / nsteps, ms_g * 1e3 / nsteps, ms_p * 1e3 / nsteps, aborted ? "ABORTED" : "ok", bad, ref[0], got[0]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
    }
    return 0;
}
