// spades_amd/csrc/smx_ingest.hip — FASTQ text -> 2-bit read batch on the device (included by smx_api.hip).
//
// The reference parses FASTQ on the CPU (kseq + zlib-ng, common/io/reads/parser.cpp) and, inside spades.py, converts reads to its
// binary format once (io::ReadConverter, common/io/reads/binary_converter.cpp:83-151): OBSERVED 6 s per 2 M reads (SURVEY.md §8a,
// row a10). One host thread parses ~1.9 M plain-FASTQ reads/s here, 100x below what the counting kernels consume, so for
// uncompressed 4-line FASTQ the raw file bytes go to HBM and are cut into reads there:
//   k_fq_count   newlines per 4 KiB block                         -> scan
//   k_fq_lines   newline j: j%4==0 starts sequence line j/4, j%4==1 ends it; checks '@' / '+' at the record and separator lines
//   k_fq_pack    one thread per read: longest ACGTacgt run, first one on ties (io::LongestValid,
//                common/io/reads/longest_valid_wrapper.hpp:16-53) + 2-bit packing (dignucl, common/sequence/nucl.hpp:132-142)
//                into a word-aligned slice of the stream (reads start at multiples of 32 nucleotides, like Sequence::BinWrite)
// Anything that is not strict 4-line FASTQ (multi-line records, FASTA, gzip) is reported as such and stays with the host parser.
#pragma once
#include "smx_device.hpp"

namespace smx {

constexpr int FQ_BLOCK = 4096;  // bytes per workgroup (16 per thread)

__global__ void __launch_bounds__(BLK) k_fq_count(const char *__restrict__ text, uint64_t n, unsigned long long *cnt) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    const uint64_t b0 = (uint64_t)blockIdx.x * FQ_BLOCK + (uint64_t)threadIdx.x * 16;
    unsigned long long c = 0;
    if (b0 + 16 <= n) {
        const uint4 v = *(const uint4 *)(text + b0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c += ((w[i] >> (8 * j)) & 0xFF) == '\n';
    } else {
        for (uint64_t p = b0; p < n; ++p) c += text[p] == '\n';
    }
    unsigned long long tot;
    block_excl_scan<unsigned long long>(c, scratch, &tot);
    if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}

// nl_virtual: the text is the end of the file and lacks the final newline; a newline is assumed at position n.
__global__ void __launch_bounds__(BLK) k_fq_lines(const char *__restrict__ text, uint64_t n, const unsigned long long *__restrict__ blk_off,
                                                  uint64_t n_records, unsigned long long *seq_start, unsigned long long *seq_end,
                                                  unsigned long long *consumed, uint32_t *bad) {
    __shared__ unsigned long long scratch[BLK / 64 + 2];
    const uint64_t b0 = (uint64_t)blockIdx.x * FQ_BLOCK + (uint64_t)threadIdx.x * 16;
    unsigned long long c = 0;
    for (uint64_t p = b0; p < b0 + 16 && p < n; ++p) c += text[p] == '\n';
    unsigned long long tot;
    unsigned long long j = blk_off[blockIdx.x] + block_excl_scan<unsigned long long>(c, scratch, &tot);
    if (blockIdx.x == 0 && threadIdx.x == 0 && n && n_records && text[0] != '@') atomicAdd(bad, 1u);
    for (uint64_t p = b0; p < b0 + 16 && p < n; ++p) {
        if (text[p] != '\n') continue;
        const uint64_t r = j >> 2;
        if (r < n_records) {
            const unsigned k = (unsigned)(j & 3);
            if (k == 0) {
                seq_start[r] = p + 1;
            } else if (k == 1) {
                seq_end[r] = (p > 0 && text[p - 1] == '\r') ? p - 1 : p;
                if (p + 1 >= n || text[p + 1] != '+') atomicAdd(bad, 1u);  // separator line
            } else if (k == 3) {
                if (p + 1 < n && r + 1 < n_records && text[p + 1] != '@') atomicAdd(bad, 1u);  // next record
                if (r + 1 == n_records) *consumed = p + 1;  // end of the last complete record
            }
        }
        ++j;
    }
}

__global__ void k_fq_words(const unsigned long long *seq_start, const unsigned long long *seq_end, uint64_t n_records,
                           unsigned long long *words, uint32_t *bad) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    const unsigned long long s = seq_start[r], e = seq_end[r];
    if (e < s || e - s > 0xFFFFFFFFull) {
        atomicAdd(bad, 1u);
        words[r] = 0;
        return;
    }
    words[r] = (e - s + 31) / 32;
}

__global__ void k_fq_pack(const char *__restrict__ text, const unsigned long long *__restrict__ seq_start,
                          const unsigned long long *__restrict__ seq_end, const unsigned long long *__restrict__ word_off, uint64_t n_records,
                          uint64_t *stream, uint64_t *start, uint32_t *len) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_records) return;
    const char *s = text + seq_start[r];
    const uint64_t n = seq_end[r] - seq_start[r];
    uint64_t *out = stream + word_off[r];
    uint64_t best_len = 0, best_pos = 0, run = 0, v = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const char c = s[i];
        const uint64_t code = (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 0;
        const bool nucl = code != 0 || c == 'A' || c == 'a';
        v |= code << ((i & 31) << 1);
        if ((i & 31) == 31) {
            out[i >> 5] = v;
            v = 0;
        }
        if (nucl) {
            ++run;
        } else {
            if (run > best_len) {
                best_len = run;
                best_pos = i - run;
            }
            run = 0;
        }
    }
    if (n & 31) out[n >> 5] = v;
    if (run > best_len) {
        best_len = run;
        best_pos = n - run;
    }
    start[r] = word_off[r] * 32 + best_pos;
    len[r] = (uint32_t)best_len;
}

}  // namespace smx
