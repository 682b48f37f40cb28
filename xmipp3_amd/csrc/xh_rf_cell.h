// xh_rf_cell.h -- the packed projection record the gridding kernel copies into LDS (written by k_rf_pack_grid*, k_rf_rowsB<PACK>).
#ifndef XH_RF_CELL_H
#define XH_RF_CELL_H
// A packed record in global memory: (re*ctf*mod*w, im*ctf*mod*w, mod*w), 12 bytes, records 12 bytes apart.  The patch copy moves 16
// bytes per record into a 16-byte LDS slot (global_load_lds_dwordx4 takes 4-byte aligned addresses: tools/ubench_dma.hip), so the
// fourth word of a slot is the next record's first word and is never used; the buffer ends with 16 spare bytes.  (XG_REC_WORDS 4: the
// round-3 records of 16 bytes, for A/B builds.)
#ifndef XG_REC_WORDS
#define XG_REC_WORDS 3
#endif
struct XgCell { float v[XG_REC_WORDS]; };
__device__ __forceinline__ void xg_put(XgCell *p, float a, float b, float c)
{
#if XG_REC_WORDS == 4
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, 0.f);
#else
    typedef float xg_v3f_ __attribute__((ext_vector_type(3)));
    typedef xg_v3f_ xg_v3u_ __attribute__((aligned(4)));
    *reinterpret_cast<xg_v3u_ *>(p) = (xg_v3f_){a, b, c};
#endif
}

#endif
