// ingest.hip — read ingestion on the device: sequence text -> the 2-bit records of reads.lib.bin.
//
// Replaces the per-base loop of SequenceManager::ReadShortReads / SequencePackage::AppendSeq (sequence_manager.cpp:109-216,
// sequence_package.h:67-69,126-129) + WriteBinarySequences (sequence_manager.cpp:375-410) for one batch of reads: the host inflates the
// file and cuts it into records (gzread + memchr: kseq.h's record rules stay where the file is), hands over the sequence characters of
// the batch back to back with their offsets, and gets the bytes of PREFIX.bin back: per read uint32 length + ceil(length / 16) words,
// base j of a word at bits 30 - 2j, zero padded, forward orientation, A0 C1 G2 T3, N -> G, anything else -> A (the table of
// formats.cpp, which stands in for the reference's uninitialised table entries).
// Bound: PCIe (1 B/base in, 0.27 B/base out); the kernel itself reads 1 B and writes 0.27 B per base once, one wave per read.
#include "common.hpp"
#include "device_utils.hpp"

namespace mgta {

__device__ __forceinline__ uint32_t base_code(uint8_t c) {
    const uint8_t l = c | 0x20;
    return l == 'c' ? 1u : (l == 'g' || l == 'n') ? 2u : l == 't' ? 3u : 0u;
}

__global__ __launch_bounds__(256) void pack_text_kernel(const uint8_t *text, const uint64_t *off, const uint64_t *out_off, uint64_t n_reads,
                                                        uint32_t *out) {
    const uint64_t waves = (uint64_t)gridDim.x * 4;
    const int lane = lane_id();
    for (uint64_t r = (uint64_t)blockIdx.x * 4 + wave_id(); r < n_reads; r += waves) {
        const uint64_t b = off[r], len = off[r + 1] - b, o = out_off[r];
        const uint64_t nw = (len + 15) / 16;
        if (lane == 0) out[o] = (uint32_t)len;
        for (uint64_t w = (uint64_t)lane; w < nw; w += 64) {
            uint32_t word = 0;
            const uint64_t p = b + 16 * w;
            const uint32_t m = (uint32_t)(len - 16 * w < 16 ? len - 16 * w : 16);
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) word = (word << 2) | (j < m ? base_code(text[p + j]) : 0u);
            out[o + 1 + w] = word;
        }
    }
}

}  // namespace mgta

using namespace mgta;

extern "C" int mgta_reads_pack_text(mgta_ctx *ctx, const char *text, uint64_t n_bytes, const uint64_t *offsets, uint64_t n_reads,
                                    uint32_t *bin_words, uint64_t capacity_words, uint64_t *n_words_out) {
    if (!ctx || !offsets || !n_words_out || (n_bytes && !text) || (capacity_words && !bin_words)) { set_error("mgta_reads_pack_text: bad argument"); return MGTA_EINVAL; }
    if (offsets[0] != 0 || offsets[n_reads] != n_bytes) { set_error("mgta_reads_pack_text: offsets must run from 0 to n_bytes"); return MGTA_EINVAL; }
    try {
        MGTA_HIP_CHECK(hipSetDevice(ctx->device));
        hipStream_t st = ctx->stream;
        std::vector<uint64_t> out_off(n_reads + 1);
        uint64_t acc = 0;
        for (uint64_t r = 0; r < n_reads; ++r) {
            if (offsets[r + 1] < offsets[r] || offsets[r + 1] - offsets[r] > 0xFFFFFFFFull) { set_error("mgta_reads_pack_text: bad offsets at read %llu", (unsigned long long)r); return MGTA_EINVAL; }
            out_off[r] = acc;
            acc += 1 + (offsets[r + 1] - offsets[r] + 15) / 16;
        }
        out_off[n_reads] = acc;
        *n_words_out = acc;
        if (acc > capacity_words) { set_error("mgta_reads_pack_text: %llu words needed, room for %llu", (unsigned long long)acc, (unsigned long long)capacity_words); return MGTA_EINVAL; }
        if (n_reads == 0) return MGTA_OK;
        DevBuf d_text, d_off, d_oo, d_out;
        d_text.alloc(n_bytes + 16, &ctx->live_bytes, &ctx->peak_bytes);
        d_off.alloc((n_reads + 1) * 8, &ctx->live_bytes, &ctx->peak_bytes);
        d_oo.alloc((n_reads + 1) * 8, &ctx->live_bytes, &ctx->peak_bytes);
        d_out.alloc(acc * 4, &ctx->live_bytes, &ctx->peak_bytes);
        if (n_bytes) MGTA_HIP_CHECK(hipMemcpyAsync(d_text.p, text, n_bytes, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(d_off.p, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, st));
        MGTA_HIP_CHECK(hipMemcpyAsync(d_oo.p, out_off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, st));
        const unsigned grid = (unsigned)std::min<uint64_t>((n_reads + 3) / 4, (uint64_t)ctx->num_cus * 32);
        hipLaunchKernelGGL(pack_text_kernel, dim3(grid), dim3(256), 0, st, d_text.as<uint8_t>(), d_off.as<uint64_t>(), d_oo.as<uint64_t>(), n_reads,
                           d_out.as<uint32_t>());
        MGTA_HIP_CHECK(hipGetLastError());
        MGTA_HIP_CHECK(hipMemcpyAsync(bin_words, d_out.p, acc * 4, hipMemcpyDeviceToHost, st));
        MGTA_HIP_CHECK(hipStreamSynchronize(st));
        return MGTA_OK;
    } catch (const HipError &e) { return e.code; }
}
