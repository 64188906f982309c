// Native ingest / output stage of the file pipeline (host only, no GPU):
//   * RIFF/WAVE decode of many files straight into one zero-padded (B, maxlen)
//     fp32 batch buffer (pinned memory supplied by the caller), multi-threaded
//     -- replaces torchaudio.load + Collate of the reference's DataLoader
//     workers (ppgs/load.py:17-30, ppgs/data/collate.py:20-27);
//   * torch.save-compatible ".pt" writer for (rows, length) fp32 tensors,
//     multi-threaded -- replaces the spawn Pool of save_masked workers
//     (ppgs/preprocess/core.py:219-221, ppgs/core.py:361-365).
// The .pt container is PyTorch's zip format (stored entries, data.pkl pickle
// protocol 2 calling torch._utils._rebuild_tensor_v2 on a FloatStorage record
// "data/0"); files written here load with torch.load(weights_only=True).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ppgs_amd.h"

namespace {

thread_local std::string t_io_error;

// ---------------------------------------------------------------- CRC-32 ----
uint32_t g_crc_table[8][256];
std::atomic<bool> g_crc_ready{false};

void crc_init() {
    if (g_crc_ready.load()) return;
    static std::atomic<bool> lock{false};
    bool expected = false;
    while (!lock.compare_exchange_weak(expected, true)) expected = false;
    if (!g_crc_ready.load()) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            g_crc_table[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s)
                g_crc_table[s][i] = (g_crc_table[s - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[s - 1][i] & 0xff];
        g_crc_ready.store(true);
    }
    lock.store(false);
}

uint32_t crc32(const void* data, size_t n) {
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {                       // slicing-by-8
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= c;
        c = g_crc_table[7][a & 0xff] ^ g_crc_table[6][(a >> 8) & 0xff] ^ g_crc_table[5][(a >> 16) & 0xff] ^
            g_crc_table[4][a >> 24] ^ g_crc_table[3][b & 0xff] ^ g_crc_table[2][(b >> 8) & 0xff] ^
            g_crc_table[1][(b >> 16) & 0xff] ^ g_crc_table[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

// ------------------------------------------------------------------- WAV ----
struct WavInfo {
    int64_t samples = 0;      // per channel
    int32_t rate = 0;
    int32_t channels = 0;
    int32_t format = 0;       // 1 PCM, 3 IEEE float
    int32_t bits = 0;
    int64_t data_offset = 0;
    int32_t block_align = 0;
};

uint32_t rd32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

int wav_parse(FILE* f, const char* path, WavInfo* info) {
    uint8_t hdr[12];
    if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4)) {
        t_io_error = std::string(path) + " is not a RIFF/WAVE file";
        return PPG_EINVAL;
    }
    bool have_fmt = false;
    for (;;) {
        uint8_t ch[8];
        if (fread(ch, 1, 8, f) != 8) {
            t_io_error = std::string(path) + " has no data chunk";
            return PPG_EINVAL;
        }
        const uint32_t size = rd32(ch + 4);
        if (!memcmp(ch, "fmt ", 4)) {
            uint8_t fmt[40] = {0};
            const size_t want = size < sizeof(fmt) ? size : sizeof(fmt);
            if (size < 16 || fread(fmt, 1, want, f) != want) {
                t_io_error = std::string(path) + ": truncated fmt chunk";
                return PPG_EINVAL;
            }
            info->format = rd16(fmt);
            info->channels = rd16(fmt + 2);
            info->rate = (int32_t)rd32(fmt + 4);
            info->block_align = rd16(fmt + 12);
            info->bits = rd16(fmt + 14);
            if (info->format == 0xFFFE && size >= 26) info->format = rd16(fmt + 24);   // WAVE_FORMAT_EXTENSIBLE
            if (fseek(f, (long)(size - want + (size & 1)), SEEK_CUR)) return PPG_EINVAL;
            have_fmt = true;
        } else if (!memcmp(ch, "data", 4)) {
            if (!have_fmt || info->block_align <= 0 || info->channels <= 0) {
                t_io_error = std::string(path) + ": data chunk before a valid fmt chunk";
                return PPG_EINVAL;
            }
            // the header is not trusted: a frame must hold all its channels' samples (the decoder
            // reads bits / 8 bytes at every block_align step), and only whole-byte sample sizes exist
            const int bits = info->bits;
            if ((bits != 8 && bits != 16 && bits != 24 && bits != 32 && bits != 64) ||
                info->block_align < info->channels * (bits / 8)) {
                char buf[128];
                snprintf(buf, sizeof(buf), ": inconsistent fmt chunk (%d channels, %d bits, block align %d)",
                         info->channels, bits, info->block_align);
                t_io_error = std::string(path) + buf;
                return PPG_EINVAL;
            }
            info->data_offset = ftell(f);
            // clamp the data chunk to what the file really holds: streaming writers leave 0 or
            // 0xFFFFFFFF in the size field, truncated files promise more than they have --
            // decode what exists, as torchaudio does
            int64_t available = 0;
            {
                const long here = ftell(f);
                if (here < 0 || fseek(f, 0, SEEK_END)) return PPG_EINVAL;
                const long end = ftell(f);
                if (end < here || fseek(f, here, SEEK_SET)) return PPG_EINVAL;
                available = (int64_t)(end - here);
            }
            int64_t bytes = (int64_t)size;
            if (size == 0 || size == 0xFFFFFFFFu || bytes > available) bytes = available;
            info->samples = bytes / info->block_align;
            return PPG_OK;
        } else if (fseek(f, (long)(size + (size & 1)), SEEK_CUR)) {
            t_io_error = std::string(path) + ": truncated chunk";
            return PPG_EINVAL;
        }
    }
}

// first channel -> fp32, scaled like the Python loader (ppgs_amd/load.py)
int wav_decode(FILE* f, const char* path, const WavInfo& w, float* dst, int64_t want) {
    const int64_t n = want < w.samples ? want : w.samples;
    std::vector<uint8_t> raw((size_t)n * w.block_align);
    if (fseek(f, (long)w.data_offset, SEEK_SET) || fread(raw.data(), 1, raw.size(), f) != raw.size()) {
        t_io_error = std::string(path) + ": truncated data chunk";
        return PPG_EINVAL;
    }
    const uint8_t* p = raw.data();
    const int step = w.block_align;
    if (w.format == 3 && w.bits == 32) {
        for (int64_t i = 0; i < n; ++i) memcpy(dst + i, p + i * step, 4);
    } else if (w.format == 3 && w.bits == 64) {
        for (int64_t i = 0; i < n; ++i) { double v; memcpy(&v, p + i * step, 8); dst[i] = (float)v; }
    } else if (w.format == 1 && w.bits == 16) {
        for (int64_t i = 0; i < n; ++i) dst[i] = (float)(int16_t)rd16(p + i * step) * (1.0f / 32768.0f);
    } else if (w.format == 1 && w.bits == 8) {
        for (int64_t i = 0; i < n; ++i) dst[i] = ((float)p[i * step] - 128.0f) * (1.0f / 128.0f);
    } else if (w.format == 1 && w.bits == 24) {
        for (int64_t i = 0; i < n; ++i) {
            const uint8_t* q = p + i * step;
            const int32_t v = (int32_t)((q[0] << 8) | (q[1] << 16) | ((uint32_t)q[2] << 24)) >> 8;
            dst[i] = (float)v * (1.0f / 8388608.0f);
        }
    } else if (w.format == 1 && w.bits == 32) {
        for (int64_t i = 0; i < n; ++i) dst[i] = (float)((double)(int32_t)rd32(p + i * step) * (1.0 / 2147483648.0));
    } else {
        char buf[96];
        snprintf(buf, sizeof(buf), ": unsupported WAV encoding (format %d, %d bits)", w.format, w.bits);
        t_io_error = std::string(path) + buf;
        return PPG_EINVAL;
    }
    return PPG_OK;
}

template <class F>
int parallel_for(int n, int threads, F fn) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    std::atomic<int> next{0};
    std::atomic<int> status{PPG_OK};
    std::string message;
    std::atomic<bool> failed{false};
    auto worker = [&] {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            const int rc = fn(i);
            if (rc != PPG_OK) {
                bool expected = false;
                if (failed.compare_exchange_strong(expected, true)) {
                    status.store(rc);
                    message = t_io_error;      // first failure wins
                }
            }
        }
    };
    if (threads == 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker);
        for (auto& th : pool) th.join();
    }
    if (failed.load()) t_io_error = message;
    return status.load();
}

// -------------------------------------------------------------- .pt writer --
void put16(std::vector<uint8_t>& b, uint16_t v) { b.push_back(v & 0xff); b.push_back(v >> 8); }
void put32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((v >> (8 * i)) & 0xff); }
void puts_(std::vector<uint8_t>& b, const char* s) { b.insert(b.end(), s, s + strlen(s)); }

struct ZipEntry { std::string name; uint32_t crc, size, offset; };

// local header (+ optional "FB" padding so the payload starts 64-byte aligned,
// as torch.save does) + payload
void zip_add(std::vector<uint8_t>& out, std::vector<ZipEntry>& dir, const std::string& name,
             const void* data, size_t size, bool align64) {
    ZipEntry e;
    e.name = name;
    e.crc = crc32(data, size);
    e.size = (uint32_t)size;
    e.offset = (uint32_t)out.size();
    size_t extra = 0;
    if (align64) {
        const size_t start = out.size() + 30 + name.size();
        extra = (64 - (start + 4) % 64) % 64 + 4;          // 4-byte extra header + padding
    }
    put32(out, 0x04034b50); put16(out, 20); put16(out, 0); put16(out, 0);      // version, flags, stored
    put16(out, 0); put16(out, 0x21);                                          // time, date (1980-01-01)
    put32(out, e.crc); put32(out, e.size); put32(out, e.size);
    put16(out, (uint16_t)name.size()); put16(out, (uint16_t)extra);
    out.insert(out.end(), name.begin(), name.end());
    if (extra) {
        out.push_back('F'); out.push_back('B'); put16(out, (uint16_t)(extra - 4));
        out.insert(out.end(), extra - 4, 'Z');
    }
    const uint8_t* p = static_cast<const uint8_t*>(data);
    out.insert(out.end(), p, p + size);
    dir.push_back(e);
}

void zip_finish(std::vector<uint8_t>& out, const std::vector<ZipEntry>& dir) {
    const uint32_t cd_offset = (uint32_t)out.size();
    for (const ZipEntry& e : dir) {
        put32(out, 0x02014b50); put16(out, 20); put16(out, 20); put16(out, 0); put16(out, 0);
        put16(out, 0); put16(out, 0x21);
        put32(out, e.crc); put32(out, e.size); put32(out, e.size);
        put16(out, (uint16_t)e.name.size()); put16(out, 0); put16(out, 0);
        put16(out, 0); put16(out, 0); put32(out, 0); put32(out, e.offset);
        out.insert(out.end(), e.name.begin(), e.name.end());
    }
    const uint32_t cd_size = (uint32_t)out.size() - cd_offset;
    put32(out, 0x06054b50); put16(out, 0); put16(out, 0);
    put16(out, (uint16_t)dir.size()); put16(out, (uint16_t)dir.size());
    put32(out, cd_size); put32(out, cd_offset); put16(out, 0);
}

// data.pkl of torch.save(tensor) for a contiguous fp32 (rows, cols) tensor
void pickle_tensor(std::vector<uint8_t>& b, int64_t rows, int64_t cols) {
    auto binint = [&](int64_t v) { b.push_back('J'); put32(b, (uint32_t)v); };
    b.push_back(0x80); b.push_back(2);
    puts_(b, "ctorch._utils\n_rebuild_tensor_v2\nq"); b.push_back(0);
    b.push_back('('); b.push_back('(');
    b.push_back('X'); put32(b, 7); puts_(b, "storage"); b.push_back('q'); b.push_back(1);
    puts_(b, "ctorch\nFloatStorage\nq"); b.push_back(2);
    b.push_back('X'); put32(b, 1); puts_(b, "0"); b.push_back('q'); b.push_back(3);
    b.push_back('X'); put32(b, 3); puts_(b, "cpu"); b.push_back('q'); b.push_back(4);
    binint(rows * cols);
    b.push_back('t'); b.push_back('q'); b.push_back(5);
    b.push_back('Q');
    binint(0);                                   // storage offset
    binint(rows); binint(cols); b.push_back(0x86); b.push_back('q'); b.push_back(6);   // size
    binint(cols); binint(1); b.push_back(0x86); b.push_back('q'); b.push_back(7);      // stride
    b.push_back(0x89);                           // requires_grad False
    puts_(b, "ccollections\nOrderedDict\nq"); b.push_back(8);
    b.push_back(')'); b.push_back('R'); b.push_back('q'); b.push_back(9);
    b.push_back('t'); b.push_back('q'); b.push_back(10);
    b.push_back('R'); b.push_back('q'); b.push_back(11);
    b.push_back('.');
}

int pt_write(const char* path, const float* src, int64_t rows, int64_t row_stride, int64_t cols) {
    if (rows * cols * 4 > 0x7fffffffLL) {
        t_io_error = std::string(path) + ": tensor too large for the 32-bit zip container";
        return PPG_EINVAL;
    }
    std::vector<float> packed((size_t)(rows * cols));
    for (int64_t r = 0; r < rows; ++r) memcpy(packed.data() + r * cols, src + r * row_stride, (size_t)cols * 4);
    std::vector<uint8_t> pkl;
    pickle_tensor(pkl, rows, cols);
    std::vector<uint8_t> out;
    out.reserve(packed.size() * 4 + 1024);
    std::vector<ZipEntry> dir;
    zip_add(out, dir, "archive/data.pkl", pkl.data(), pkl.size(), false);
    zip_add(out, dir, "archive/byteorder", "little", 6, false);
    zip_add(out, dir, "archive/data/0", packed.data(), packed.size() * 4, true);
    zip_add(out, dir, "archive/version", "3\n", 2, false);
    zip_finish(out, dir);
    // write next to the target and rename: a reader never sees a half-written file, and a
    // failed flush (disk full) is reported instead of leaving a truncated .pt behind
    const std::string tmp = std::string(path) + ".tmp~";
    FILE* f = fopen(tmp.c_str(), "wb");
    const bool written = f && fwrite(out.data(), 1, out.size(), f) == out.size();
    const bool closed = f ? fclose(f) == 0 : false;
    if (!written || !closed || rename(tmp.c_str(), path) != 0) {
        (void)remove(tmp.c_str());
        t_io_error = std::string("cannot write ") + path;
        return PPG_EINVAL;
    }
    return PPG_OK;
}

}  // namespace

extern "C" {

const char* ppg_io_last_error(void) { return t_io_error.c_str(); }

int ppg_wav_info(const char* path, int64_t* samples, int32_t* sample_rate, int32_t* channels) {
    if (!path) return PPG_EINVAL;
    FILE* f = fopen(path, "rb");
    if (!f) { t_io_error = std::string("cannot open ") + path; return PPG_EINVAL; }
    WavInfo w;
    const int rc = wav_parse(f, path, &w);
    fclose(f);
    if (rc) return rc;
    if (samples) *samples = w.samples;
    if (sample_rate) *sample_rate = w.rate;
    if (channels) *channels = w.channels;
    return PPG_OK;
}

int ppg_wav_read_batch(const char* const* paths, int count, float* dst, int64_t row_stride,
                       int64_t max_samples, int64_t* samples_out, int32_t* rates_out, int threads) {
    if (!paths || !dst || count <= 0 || max_samples > row_stride) {
        t_io_error = "ppg_wav_read_batch: bad argument";
        return PPG_EINVAL;
    }
    return parallel_for(count, threads, [&](int i) -> int {
        FILE* f = fopen(paths[i], "rb");
        if (!f) { t_io_error = std::string("cannot open ") + paths[i]; return PPG_EINVAL; }
        WavInfo w;
        int rc = wav_parse(f, paths[i], &w);
        float* row = dst + (size_t)i * row_stride;
        int64_t n = 0;
        if (!rc) {
            n = w.samples < max_samples ? w.samples : max_samples;
            rc = wav_decode(f, paths[i], w, row, n);
        }
        fclose(f);
        if (rc) return rc;
        memset(row + n, 0, (size_t)(row_stride - n) * sizeof(float));      // collate zero padding
        if (samples_out) samples_out[i] = w.samples;
        if (rates_out) rates_out[i] = w.rate;
        return PPG_OK;
    });
}

int ppg_pt_write_batch(const char* const* paths, int count, const float* src, int64_t item_stride,
                       int rows, int64_t row_stride, const int64_t* cols, int threads) {
    if (!paths || !src || !cols || count <= 0 || rows <= 0) {
        t_io_error = "ppg_pt_write_batch: bad argument";
        return PPG_EINVAL;
    }
    crc_init();
    return parallel_for(count, threads, [&](int i) -> int {
        if (cols[i] < 0 || cols[i] > row_stride) {
            t_io_error = std::string(paths[i]) + ": length outside the source row";
            return PPG_EINVAL;
        }
        return pt_write(paths[i], src + (size_t)i * item_stride, rows, row_stride, cols[i]);
    });
}

}  // extern "C"
