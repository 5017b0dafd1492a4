// What does a fork (side stream made to wait for the main stream's progress) cost the MAIN stream?
// main: N x [K_main (~15 us)], after each a fork releases [K_side (~25 us)] on a second stream.
//   mode 0: no fork at all (main alone)            mode 1: hipEventRecord + hipStreamWaitEvent
//   mode 2: hipStreamWriteValue32 + hipStreamWaitValue32 (memory flag instead of an event)
//   mode 3: the event bound to the kernel itself (hipExtLaunchKernelGGL's stopEvent: no marker packet of its own) + hipStreamWaitEvent
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float *out, int iters) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 123.456f) out[0] = v;
}

int main() {
    hipStream_t m, s;
    hipStreamCreateWithFlags(&m, hipStreamNonBlocking); hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float *buf; hipMalloc(&buf, 4096);
    uint32_t *flag; hipMalloc(&flag, 64); hipMemset(flag, 0, 64);
    const int N = 40;
    std::vector<hipEvent_t> ev(N);
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    const int main_iters = 6000, side_iters = 10000;   // calibrated below
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            hipMemsetAsync(flag, 0, 4, m);
            hipDeviceSynchronize();
            hipEventRecord(t0, m);
            for (int i = 0; i < N; ++i) {
                if (mode == 4) hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, m, nullptr, ev[i], 0, buf, main_iters);   // stop event on every launch, nobody waits
                else if (mode == 3) { hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, m, nullptr, ev[i], 0, buf, main_iters); hipStreamWaitEvent(s, ev[i], 0); }
                else spin_kernel<<<256, 256, 0, m>>>(buf, main_iters);
                if (mode == 1) { hipEventRecord(ev[i], m); hipStreamWaitEvent(s, ev[i], 0); }
                if (mode == 2) { hipStreamWriteValue32(m, flag, (uint32_t)(i + 1), 0); hipStreamWaitValue32(s, flag, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xFFFFFFFFu); }
                if (mode && mode != 4) spin_kernel<<<64, 256, 0, s>>>(buf + 512, side_iters);
            }
            hipEventRecord(t1, m);
            hipEventSynchronize(t1); hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, t0, t1);
            if (ms < best) best = ms;
        }
        printf("mode %d (%s): main stream %.2f us per iteration\n", mode,
               mode == 0 ? "no fork" : (mode == 1 ? "event record + stream wait event" : (mode == 2 ? "stream write value + stream wait value" : (mode == 3 ? "stop event of the kernel launch + stream wait event" : "stop event on every launch, no side stream"))), best * 1000.f / N);
    }
    return 0;
}
