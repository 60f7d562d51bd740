// spades_amd/csrc/smx_gfa.hip — the GFA text of a device-resident graph, formatted on the device (included by smx_api.hip).
//
// GFAWriter::WriteSegmentsAndLinks (io/graph/gfa_writer.cpp:19-47,73-87,113-116):
//   S <id> <sequence> DP:f:<float(cov)> KC:i:<raw>      one line per canonical edge, id = 3 + 2 i
//   L <e1> <+|-> <e2> <+|-> <k>M                        per vertex (id order): every incoming edge x every outgoing edge
// The host writer (smx_graph_host.hpp write_gfa) needs the whole graph unpacked in host memory first — at 20 M reads 3.9 s of the
// tool's 6.9 s for 5 GB of text. Here the lines get their lengths, a prefix sum places them, and the bytes are written straight into
// one device buffer (2 bits -> ASCII, decimal ids) that goes to the file through page-locked buffers and several pwrite threads. Only
// the "DP:f:%g" fields of a coverage run are formatted on the host (the reference prints a float through an ostream: %g), as ready
// tag strings that the device copies.
#pragma once
#include "smx_device.hpp"

namespace smx {

__device__ __forceinline__ uint32_t gfa_digits(uint64_t v) {
    uint32_t d = 1;
    while (v >= 10) {
        v /= 10;
        ++d;
    }
    return d;
}
__device__ __forceinline__ void gfa_put_dec(char *dst, uint64_t v, uint32_t nd) {
    for (uint32_t i = nd; i-- > 0;) {
        dst[i] = (char)('0' + (uint32_t)(v % 10));
        v /= 10;
    }
}
constexpr uint32_t GFA_PLAIN_TAG = 15;  // "\tDP:f:0\tKC:i:0\n"

// bytes of every S line
__global__ void k_gfa_s_len(const unsigned long long *__restrict__ elen, uint64_t ne, const uint32_t *__restrict__ taglen, unsigned long long *slen) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (uint64_t)gridDim.x * blockDim.x)
        slen[i] = 2 + gfa_digits(3 + 2 * i) + 1 + elen[i] + (taglen ? taglen[i] : GFA_PLAIN_TAG);
}
// one wave per S line: lane 0 writes the head, all lanes the sequence (64 consecutive bytes per step), then the tag
__global__ void __launch_bounds__(256) k_gfa_s_write(const uint64_t *__restrict__ words, const unsigned long long *__restrict__ eoffw,
                                                     const unsigned long long *__restrict__ elen, uint64_t ne, const unsigned long long *__restrict__ soff,
                                                     const char *__restrict__ tagpool, const unsigned long long *__restrict__ tagoff, char *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t i = wave; i < ne; i += nwaves) {
        char *p = out + soff[i];
        const uint32_t nd = gfa_digits(3 + 2 * i);
        if (lane == 0) {
            p[0] = 'S';
            p[1] = '\t';
            gfa_put_dec(p + 2, 3 + 2 * i, nd);
            p[2 + nd] = '\t';
        }
        p += 3 + nd;
        const uint64_t *w = words + eoffw[i];
        const uint64_t len = elen[i];
        for (uint64_t t = lane; t < len; t += 64) p[t] = "ACGT"[(w[t >> 5] >> ((t & 31u) << 1)) & 3u];
        p += len;
        if (tagpool) {
            const unsigned long long a = tagoff[i], n = tagoff[i + 1] - a;
            for (uint64_t t = lane; t < n; t += 64) p[t] = tagpool[a + t];
        } else if (lane < GFA_PLAIN_TAG) {
            p[lane] = "\tDP:f:0\tKC:i:0\n"[lane];
        }
    }
}

// out-edge lists of v and conj(v) of vertex number vn, sorted by edge id (smx_graph_host.hpp vertex_edges; link records as k_link_keys
// left them: w[0] >> sh = the vertex's key, w[1] = edge << 2 | is_rc << 1 | is_start)
__device__ __forceinline__ void gfa_vertex_edges(const Rec<2> *__restrict__ lr, uint64_t nlrec, unsigned sh, const unsigned long long *__restrict__ vstart, uint64_t vn,
                                                 const uint8_t *__restrict__ eself, uint64_t *outv, uint32_t &no, uint64_t *outc, uint32_t &nc) {
    no = nc = 0;
    const uint64_t i0 = vstart[vn], h = lr[i0].w[0] >> sh;
    for (uint64_t j = i0; j < nlrec && (lr[j].w[0] >> sh) == h && no < 8 && nc < 8; ++j) {
        const uint64_t em = lr[j].w[1], e = em >> 2;
        const uint64_t ce = eself[(e - 3) >> 1] ? e : e + 1;
        const bool is_rc = (em >> 1) & 1, is_start = em & 1;
        if (is_start) {
            if (!is_rc) outv[no++] = e;
            else outc[nc++] = e;
        } else {
            if (!is_rc) outc[nc++] = ce;
            else outv[no++] = ce;
        }
    }
    for (uint32_t a = 1; a < no; ++a)  // (at most 8 entries: insertion sort)
        for (uint32_t b = a; b > 0 && outv[b - 1] > outv[b]; --b) {
            const uint64_t t = outv[b];
            outv[b] = outv[b - 1];
            outv[b - 1] = t;
        }
    for (uint32_t a = 1; a < nc; ++a)
        for (uint32_t b = a; b > 0 && outc[b - 1] > outc[b]; --b) {
            const uint64_t t = outc[b];
            outc[b] = outc[b - 1];
            outc[b - 1] = t;
        }
}
// WRITE = false: bytes (llen) and links (nlinks) of every vertex; WRITE = true: the lines at loff[vn]
template <bool WRITE>
__global__ void __launch_bounds__(256) k_gfa_l(const Rec<2> *__restrict__ lr, uint64_t nlrec, unsigned sh, const unsigned long long *__restrict__ vstart, uint64_t nv,
                                               const uint8_t *__restrict__ eself, unsigned k, unsigned long long *llen, unsigned long long *nlinks,
                                               const unsigned long long *__restrict__ loff, char *out) {
    const uint32_t kd = gfa_digits(k);
    unsigned long long links = 0;
    for (uint64_t vn = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; vn < nv; vn += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t outv[8], outc[8];
        uint32_t no, nc;
        gfa_vertex_edges(lr, nlrec, sh, vstart, vn, eself, outv, no, outc, nc);
        unsigned long long bytes = 0;
        char *p = WRITE ? out + loff[vn] : nullptr;
        for (uint32_t a = 0; a < nc; ++a) {
            const uint64_t oc = outc[a];
            const uint64_t inc = eself[(oc - 3) >> 1] ? oc : (((oc - 3) & 1) ? oc - 1 : oc + 1);
            const uint64_t cin = 3 + (((inc - 3) >> 1) << 1);
            const uint32_t d1 = gfa_digits(cin);
            for (uint32_t c = 0; c < no; ++c) {
                const uint64_t oe = outv[c], cout = 3 + (((oe - 3) >> 1) << 1);
                const uint32_t d2 = gfa_digits(cout);
                if constexpr (WRITE) {
                    *p++ = 'L';
                    *p++ = '\t';
                    gfa_put_dec(p, cin, d1);
                    p += d1;
                    *p++ = '\t';
                    *p++ = inc == cin ? '+' : '-';
                    *p++ = '\t';
                    gfa_put_dec(p, cout, d2);
                    p += d2;
                    *p++ = '\t';
                    *p++ = oe == cout ? '+' : '-';
                    *p++ = '\t';
                    gfa_put_dec(p, k, kd);
                    p += kd;
                    *p++ = 'M';
                    *p++ = '\n';
                }
                bytes += 2 + d1 + 3 + d2 + 3 + kd + 2;
                ++links;
            }
        }
        if constexpr (!WRITE) llen[vn] = bytes;
    }
    if constexpr (!WRITE) {
        for (int o = 32; o > 0; o >>= 1) links += __shfl_down(links, o, 64);
        if ((threadIdx.x & 63) == 0 && links) atomicAdd(nlinks, links);
    }
}

}  // namespace smx
