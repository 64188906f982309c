// Sample-rate conversion on the device: the windowed-sinc polyphase resampler of
// torchaudio.transforms.Resample with its defaults (Hann window,
// lowpass_filter_width 6, rolloff 0.99), which reference ppgs/core.py:599-608
// applies to audio that is not at 16 kHz.  torchaudio is a third-party
// dependency absent here; the algorithm is restated from its published form
// (same restatement as oracle/ppg_oracle.py::resample, which the tests compare
// against).
//
//   orig, new = rates / gcd;  base = min(orig, new) * 0.99;  width = ceil(6 * orig / base)
//   kernel[i][k] = sinc(pi t) * cos^2(pi t / 12) * base / orig,
//       t = clamp((-i / new + (k - width) / orig) * base, -6, 6),   k < 2 width + orig
//   out[j new + i] = sum_k x[j orig + k - width] * kernel[i][k]     (x = 0 outside)
//   truncated to ceil(new * samples / orig) samples.
// Memory-trivial (a second of 48 kHz audio is 192 KB): one thread per output
// sample, the filter bank (new x (2 width + orig) floats, built in double on the
// host, cached per device and rate pair) is read through the caches.
#include "../../include/ppgs_amd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <numeric>
#include <utility>
#include <vector>

namespace ppg {
int fail_message(int code, const char* fmt, ...);   // ppg_engine.hip: sets ppg_last_error()
}

namespace {

__global__ __launch_bounds__(256) void resample_kernel(
    const float* __restrict__ audio, long long samples, const float* __restrict__ bank, int orig, int now,
    int width, int taps, float* __restrict__ out, long long out_samples)
{
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= out_samples) return;
    const float* x = audio + (size_t)blockIdx.y * samples;
    const long long j = n / now;
    const int i = (int)(n - j * now);
    const float* h = bank + (size_t)i * taps;
    const long long first = j * orig - width;          // source index of tap 0
    float acc = 0.f;
    for (int k = 0; k < taps; ++k) {
        const long long p = first + k;
        if (p >= 0 && p < samples) acc = fmaf(x[p], h[k], acc);
    }
    out[(size_t)blockIdx.y * out_samples + n] = acc;
}

struct Bank {
    float* device = nullptr;
    int orig = 0, now = 0, width = 0, taps = 0;
};
std::mutex g_mu;
std::map<std::pair<int, std::pair<int, int>>, Bank> g_banks;

}  // namespace

extern "C" {

int64_t ppg_resample_length(int64_t samples, int orig_rate, int new_rate) {
    if (samples < 0 || orig_rate <= 0 || new_rate <= 0) return -1;
    const int g = std::gcd(orig_rate, new_rate);
    const int64_t orig = orig_rate / g, now = new_rate / g;
    return (now * samples + orig - 1) / orig;
}

int ppg_resample(int device, const float* audio, int batch, int64_t samples, int orig_rate, int new_rate,
                 float* out, void* stream) {
    if (!audio || !out || batch <= 0 || samples <= 0 || orig_rate <= 0 || new_rate <= 0)
        return ppg::fail_message(PPG_EINVAL, "resample: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return ppg::fail_message(PPG_EDEVICE, "no HIP device: the resampler has no CPU path");
    if (hipSetDevice(device) != hipSuccess) return ppg::fail_message(PPG_EDEVICE, "hipSetDevice(%d) failed", device);
    const int g = std::gcd(orig_rate, new_rate);
    const int orig = orig_rate / g, now = new_rate / g;
    Bank bank;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        Bank& b = g_banks[{device, {orig, now}}];
        if (!b.device) {
            const double lowpass = 6.0, rolloff = 0.99;
            const double base = std::min(orig, now) * rolloff;
            const int width = (int)std::ceil(lowpass * orig / base);
            const int taps = 2 * width + orig;
            std::vector<float> host((size_t)now * taps);
            for (int i = 0; i < now; ++i)
                for (int k = 0; k < taps; ++k) {
                    double t = (-(double)i / now + (double)(k - width) / orig) * base;
                    t = std::min(std::max(t, -lowpass), lowpass);
                    const double c = std::cos(t * M_PI / lowpass / 2.0);
                    const double a = t * M_PI;
                    const double sinc = a == 0.0 ? 1.0 : std::sin(a) / a;
                    host[(size_t)i * taps + k] = (float)(sinc * c * c * (base / orig));
                }
            float* d = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&d), host.size() * sizeof(float)) != hipSuccess ||
                hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
                return ppg::fail_message(PPG_EDEVICE, "resample: filter bank upload failed");
            b.device = d; b.orig = orig; b.now = now; b.width = width; b.taps = taps;
        }
        bank = b;
    }
    const int64_t out_samples = ppg_resample_length(samples, orig_rate, new_rate);
    const dim3 grid((unsigned)((out_samples + 255) / 256), (unsigned)batch);
    hipLaunchKernelGGL(resample_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), audio,
                       (long long)samples, bank.device, orig, now, bank.width, bank.taps, out, (long long)out_samples);
    const hipError_t he = hipGetLastError();
    if (he != hipSuccess) return ppg::fail_message(PPG_EDEVICE, "resample: %s", hipGetErrorString(he));
    return PPG_OK;
}

}  // extern "C"
