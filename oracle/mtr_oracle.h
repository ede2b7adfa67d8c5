/*
 * mtr_oracle.h — CPU ORACLE interface (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See mtr_oracle.c for the reference file:line each function restates.
 * PARITY UNPINNED against real Mitsuba (see mtr_oracle.c header).
 */
#ifndef MTR_ORACLE_H
#define MTR_ORACLE_H
#include "../include/mitransient_amd.h"   /* POD scene / params / counters structs of the boundary */
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_splat_rec {
    uint32_t lane, depth_kind /* depth | kind<<16 (0 = emission, 1 = emitter sampling) */, pixel, bin;
    float r, g, b, opl;
} orc_splat_rec;

int  orc_render(const mtr_scene_desc *d, const mtr_render_params *P, float *transient_hwt4, float *steady_hw4,
                mtr_counters *out, int n_threads, int use_bvh,
                orc_splat_rec *log, uint64_t log_cap, uint64_t *log_n);
void orc_develop(const mtr_film_desc *f, const float *transient_hwt4, float *transient_hwt3,
                 const float *steady_hw4, float *steady_hw3);
void orc_phasor_term(float freq, float opl, float *c, float *s);
void orc_splat_add(const mtr_film_desc *f, uint64_t n, const uint32_t *pixel, const float *opl,
                   const float *r, const float *g, const float *b, float *transient_hwt4,
                   const uint32_t *laser_x, const uint32_t *laser_y);
int  orc_bin_index(float distance, float start_opl, float bin_width_opl, uint32_t T);
void orc_intersect(const mtr_scene_desc *d, uint32_t n, const float *o3, const float *d3, const float *maxt,
                   int use_bvh, float *t_out, int32_t *prim_out, uint8_t *occluded_out);
void orc_camera_ray(const mtr_scene_desc *d, uint32_t px, uint32_t py, float j1, float j2, float *o3, float *d3, float *maxt);
void orc_pcg32_stream(uint64_t initstate, uint64_t initseq, uint32_t n, uint32_t *out_u32, float *out_f32);
void orc_sampler_stream(uint32_t seed_value, uint32_t lane, uint32_t n, float *out);
void orc_tea32(uint32_t v0, uint32_t v1, int rounds, uint32_t *out2);
void orc_sincos_q(float x, float *s, float *c);
void orc_square_to_cos_hemi(float u1, float u2, float *out3);
int  orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
