// Host side of the whole-image split-product MBConv kernel (mbconv_img_x3.hpp): weight packing and the dispatch over
// EfficientNet-B0's layer shapes.
#include "mbconv_img_x3.hpp"

namespace icap {

// chunk width of the whole-image split-product kernel for this resolution: 64 channels at 14 x 14, 32 at 7 x 7 (192 input
// channels: the stationary A fragments take 288 registers, the accumulators of a second column block would not fit)
static int x3_chunk_width(int H) { return H == 14 ? 64 : 32; }

// bytes of the packed expand weights the whole-image split-product kernel reads (0: this layer shape has no such kernel)
long long expand_dwconv_img_x3_we_bytes(int H, int W, int Cin, int Cexp, int KS) {
    extern int expand_dwconv_whole_image(int H, int W, int Cin, int Cexp, int KS, int stride);
    if (!expand_dwconv_whole_image(H, W, Cin, Cexp, KS, 1)) return 0;
    const int xch = x3_chunk_width(H);
    const long long nchunk = (Cexp + xch - 1) / xch;
    return nchunk * (Cin / 16) * 96LL * xch;
}

int expand_dwconv_img_x3_pack(const float* we, void* out, int H, int W, int Cin, int Cexp, int KS, hipStream_t st) {
    if (!we || !out || expand_dwconv_img_x3_we_bytes(H, W, Cin, Cexp, KS) == 0) {
        set_error("expand_dwconv (whole image, split products): no such kernel for %dx%d Cin=%d Cexp=%d %dx%d", H, W, Cin, Cexp, KS, KS);
        return ICAP_ERR_ARG;
    }
    if (!aligned16(out)) { set_error("expand_dwconv (whole image, split products): packed weights must be 16-byte aligned"); return ICAP_ERR_ALIGN; }
    const int xch = x3_chunk_width(H), nchunk = (Cexp + xch - 1) / xch;
    const long long total = (long long)nchunk * (Cin / 16) * 2 * xch;
    hipLaunchKernelGGL(x3_pack_we_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, we, reinterpret_cast<uint4*>(out), Cexp, Cin,
                       xch, nchunk);
    return launch_status("expand_dwconv pack");
}

// stride 2 (14 x 14 -> 7 x 7, TF-'same' padding: one row / column of zeros before the image for a 5 x 5 filter): the layer
// shapes that have a kernel
int expand_dwconv_img_x3_s2_supported(int H, int W, int Cin, int Cexp, int KS, int pad_t, int pad_l) {
    return H == 14 && W == 14 && Cin == 112 && KS == 5 && pad_t == 1 && pad_l == 1 && Cexp > 0 && Cexp % 4 == 0;
}

static int expand_dwconv_img_x3_launch(const float* x, const void* we3p, const float* es, const float* eb, const float* wd, const float* ds,
                                       const float* db, float* y, float* pooled, int B, int H, int W, int Cin, int Cexp, int KS, int pad_t,
                                       int pad_l, int stride, hipStream_t st) {
    const bool shape_ok = stride == 2 ? expand_dwconv_img_x3_s2_supported(H, W, Cin, Cexp, KS, pad_t, pad_l) != 0
                                      : (expand_dwconv_img_x3_we_bytes(H, W, Cin, Cexp, KS) != 0 && pad_t == KS / 2 && pad_l == KS / 2);
    if (!x || !we3p || !wd || !y || B <= 0 || !shape_ok) {
        set_error("expand_dwconv (whole image, split products): unsupported layer %dx%d Cin=%d %dx%d pad %d,%d", H, W, Cin, KS, KS, pad_t, pad_l);
        return ICAP_ERR_ARG;
    }
    if (!aligned16(x) || !aligned16(we3p) || !aligned16(wd) || !aligned16(y) || (ds && !aligned16(ds)) || (db && !aligned16(db))) {
        set_error("expand_dwconv (whole image, split products): pointers must be 16-byte aligned");
        return ICAP_ERR_ALIGN;
    }
    const int imgs = H == 14 ? 1 : 4;
    const int xch = x3_chunk_width(H), nchunk = (Cexp + xch - 1) / xch;
    const long long groups = (B + imgs - 1) / imgs;
    // one workgroup per CU is resident: split the chunk list while that leaves fewer than two rounds of workgroups
    int nsplit = 1;
    while (groups * nsplit < 512 && nsplit * 2 <= nchunk / 2) nsplit *= 2;
    const int cpg = (nchunk + nsplit - 1) / nsplit;
    nsplit = (nchunk + cpg - 1) / cpg;
    const long long wgs = groups * nsplit;
    if (wgs > 0x7fffffffLL) { set_error("expand_dwconv (whole image, split products): problem too large"); return ICAP_ERR_LIMIT; }
    const dim3 grid((unsigned)wgs), block(256), block8(512);
    const uint4* wp = reinterpret_cast<const uint4*>(we3p);
#define ICAP_X3(CIN_, KS_, HW_, NCB_) \
    hipLaunchKernelGGL((expand_dw_img_x3_kernel<CIN_, KS_, HW_, NCB_>), grid, block, 0, st, x, wp, es, eb, wd, ds, db, y, pooled, B, Cexp, nchunk, cpg, nsplit)
    // 14 x 14: eight waves, the A planes of the first five K slices in registers (all of them for 80 input channels)
#define ICAP_X3W8(CIN_, KS_) \
    hipLaunchKernelGGL((expand_dw_img14_w8_kernel<CIN_, KS_, 5>), grid, block8, 0, st, x, wp, es, eb, wd, ds, db, y, pooled, B, Cexp, nchunk, cpg, nsplit)
    if (stride == 2)
        hipLaunchKernelGGL((expand_dw_img14_w8_kernel<112, 5, 5, 2>), grid, block8, 0, st, x, wp, es, eb, wd, ds, db, y, pooled, B, Cexp, nchunk,
                           cpg, nsplit);
    else if (Cin == 80 && KS == 3) ICAP_X3W8(80, 3);
    else if (Cin == 80) ICAP_X3W8(80, 5);
    else if (Cin == 112 && KS == 3) ICAP_X3W8(112, 3);
    else if (Cin == 112) ICAP_X3W8(112, 5);
    else if (KS == 3) ICAP_X3(192, 3, 7, 1);
    else ICAP_X3(192, 5, 7, 1);
#undef ICAP_X3W8
#undef ICAP_X3
    return launch_status("expand_dwconv (whole image, split products)");
}

int expand_dwconv_img_x3_f32(const float* x, const void* we3p, const float* es, const float* eb, const float* wd, const float* ds,
                             const float* db, float* y, float* pooled, int B, int H, int W, int Cin, int Cexp, int KS, int pad_t,
                             int pad_l, hipStream_t st) {
    return expand_dwconv_img_x3_launch(x, we3p, es, eb, wd, ds, db, y, pooled, B, H, W, Cin, Cexp, KS, pad_t, pad_l, 1, st);
}

// y [B][7][7][Cexp]; the packed weights are those of the stride-1 layer of the same shape (expand_dwconv_img_x3_pack)
int expand_dwconv_img_x3_s2_f32(const float* x, const void* we3p, const float* es, const float* eb, const float* wd, const float* ds,
                                const float* db, float* y, float* pooled, int B, int H, int W, int Cin, int Cexp, int KS, int pad_t,
                                int pad_l, hipStream_t st) {
    return expand_dwconv_img_x3_launch(x, we3p, es, eb, wd, ds, db, y, pooled, B, H, W, Cin, Cexp, KS, pad_t, pad_l, 2, st);
}

}  // namespace icap
