// Test-only entry to ppg_attn64.hip's launcher (and, for A/B, nothing else): tests/test_gpu_attn64.py packs random
// Q | K rows and V^T in the engine's layouts with torch, launches the kernel through this and compares the attention
// output with softmax(q k^T) v computed by torch from the same 16-bit values.  Not part of the product library.
#include "../../ppgs_amd/csrc/ppg_attn64.hip"

extern "C" int attn64_probe_launch(int precision, const void* qk, int qk_ld_bytes, const void* vt, int vt_ld_bytes, void* ao,
                                   int H, int causal, const void* items, int nitems, int heads, int M, int ao_tiled, void* stream) {
    AttnArgs a{};
    a.qk = static_cast<const char*>(qk); a.qk_ld_bytes = qk_ld_bytes;
    a.vt = static_cast<const char*>(vt); a.vt_ld_bytes = vt_ld_bytes;
    a.ao = static_cast<char*>(ao); a.H = H; a.causal = causal;
    a.items = static_cast<const AttnItem*>(items); a.win = nullptr; a.M = M; a.ao_tiled = ao_tiled; a.heads = heads;
    return (int)ppg::launch_attn64(precision, a, nitems, heads, static_cast<hipStream_t>(stream));
}
