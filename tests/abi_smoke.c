/* abi_smoke.c — the drop-in boundary used from plain C: nothing but include/mitransient_amd.h, the HIP runtime's C API
 * for device memory, and libmitransient_amd.so.  A diffuse floor under a small area light (one analytic rectangle each),
 * 16 x 16 pixels, 32 time bins, 8 samples: render, develop, copy back, check the energy identity
 * sum_t transient == steady (the window covers every path) and the counters.
 * Exit code 0 = pass; 3 = no HIP device (the library has no CPU path and says so); anything else = failure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <hip/hip_runtime_api.h>
#include "../include/mitransient_amd.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != MTR_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mtr_last_error(ctx)); return 1; } } while (0)

static void rect(float *v, const float c[3], const float du[3], const float dv[3])
{   /* corners (-1,-1) (1,-1) (1,1) (-1,1) -> triangles (0,1,2) (0,2,3) */
    float p[4][3]; const int sx[4] = { -1, 1, 1, -1 }, sy[4] = { -1, -1, 1, 1 };
    for (int k = 0; k < 4; ++k) for (int a = 0; a < 3; ++a) p[k][a] = c[a] + sx[k] * du[a] + sy[k] * dv[a];
    const int idx[6] = { 0, 1, 2, 0, 2, 3 };
    for (int k = 0; k < 6; ++k) memcpy(v + 3 * k, p[idx[k]], 3 * sizeof(float));
}

int main(void)
{
    mtr_ctx *ctx = NULL;
    int rc = mtr_ctx_create(0, &ctx);
    if (rc == MTR_ERR_NO_DEVICE) { fprintf(stderr, "no device: %s\n", mtr_last_error(NULL)); return 3; }
    if (rc != MTR_OK) { fprintf(stderr, "mtr_ctx_create -> %d: %s\n", rc, mtr_last_error(NULL)); return 1; }

    enum { W = 16, H = 16, T = 32, SPP = 8 };
    float verts[4 * 9];
    const float fc[3] = { 0, 0, 0 }, fdu[3] = { 2, 0, 0 }, fdv[3] = { 0, 0, -2 };          /* floor, normal +y */
    const float lc[3] = { 0, 2, 0 }, ldu[3] = { 0.5f, 0, 0 }, ldv[3] = { 0, 0, 0.5f };     /* light, normal -y */
    rect(verts, fc, fdu, fdv); rect(verts + 18, lc, ldu, ldv);
    uint32_t tri_mat[4] = { 0, 0, 0, 0 }; int32_t tri_em[4] = { -1, -1, 0, 0 };
    mtr_material mat; memset(&mat, 0, sizeof mat);
    mat.type = MTR_BSDF_DIFFUSE; mat.a[0] = mat.a[1] = mat.a[2] = 0.5f; mat.int_ior = mat.ext_ior = 1.0f;
    mtr_emitter em; memset(&em, 0, sizeof em);
    memcpy(em.center, lc, sizeof lc); memcpy(em.du, ldu, sizeof ldu); memcpy(em.dv, ldv, sizeof ldv);
    em.radiance[0] = em.radiance[1] = em.radiance[2] = 10.0f;
    mtr_shape shapes[2]; memset(shapes, 0, sizeof shapes);
    shapes[0].first_tri = 0; shapes[0].n_tris = 2; shapes[0].is_rectangle = 1;
    memcpy(shapes[0].center, fc, sizeof fc); memcpy(shapes[0].du, fdu, sizeof fdu); memcpy(shapes[0].dv, fdv, sizeof fdv);
    shapes[1].first_tri = 2; shapes[1].n_tris = 2; shapes[1].is_rectangle = 1;
    memcpy(shapes[1].center, lc, sizeof lc); memcpy(shapes[1].du, ldu, sizeof ldu); memcpy(shapes[1].dv, ldv, sizeof ldv);

    mtr_scene_desc d; memset(&d, 0, sizeof d);
    d.n_tris = 4; d.tri_verts = verts; d.tri_material = tri_mat; d.tri_emitter = tri_em;
    d.n_materials = 1; d.materials = &mat; d.n_emitters = 1; d.emitters = &em;
    d.n_shapes = 2; d.shapes = shapes;
    /* camera at (0, 1, 4) looking along -z: sample (sx, sy) in [0,1]^2 -> a point of the z = 1 plane of the camera frame */
    const float half = 0.4f;
    const float s2c[16] = { -2 * half, 0, 0, half,   0, -2 * half, 0, half,   0, 0, 0, 1,   0, 0, 0, 1 };
    const float c2w[16] = { -1, 0, 0, 0,   0, 1, 0, 1,   0, 0, -1, 4,   0, 0, 0, 1 };
    memcpy(d.camera.sample_to_camera, s2c, sizeof s2c); memcpy(d.camera.to_world, c2w, sizeof c2w);
    d.camera.near_clip = 0.01f; d.camera.far_clip = 100.0f;
    d.film.width = d.film.crop_width = W; d.film.height = d.film.crop_height = H;
    d.film.temporal_bins = T; d.film.start_opl = 0.0f; d.film.bin_width_opl = 2.0f;       /* 0 .. 64: every path */

    mtr_scene *scene = NULL;
    CHECK(mtr_scene_create(ctx, &d, &scene));
    float *t4 = NULL, *s4 = NULL, *t3 = NULL, *s3 = NULL;
    const size_t nt = (size_t)W * H * T, np = (size_t)W * H;
    if (hipMalloc((void **)&t4, nt * 16) || hipMalloc((void **)&s4, np * 16) || hipMalloc((void **)&t3, nt * 12) ||
        hipMalloc((void **)&s3, np * 12)) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    CHECK(mtr_film_clear(ctx, &d.film, t4, s4));
    mtr_render_params p; memset(&p, 0, sizeof p);
    p.spp_total = SPP; p.spp_begin = 0; p.spp_end = SPP; p.pixel_begin = 0; p.pixel_end = W * H;
    p.seed = 0; p.max_depth = 4; p.rr_depth = 5; p.flags = MTR_FLAG_FILM_ZERO; p.mode = MTR_MODE_AUTO;
    mtr_counters cnt;
    CHECK(mtr_render(scene, &p, t4, s4, &cnt, NULL));
    CHECK(mtr_film_develop(ctx, &d.film, t4, t3, s4, s3));
    float *ht = (float *)malloc(nt * 12), *hs = (float *)malloc(np * 12);
    if (hipDeviceSynchronize() || hipMemcpy(ht, t3, nt * 12, hipMemcpyDeviceToHost) || hipMemcpy(hs, s3, np * 12, hipMemcpyDeviceToHost)) {
        fprintf(stderr, "copy back failed\n"); return 1; }
    double et = 0, es = 0, diff = 0;
    for (size_t i = 0; i < np; ++i)
        for (int c = 0; c < 3; ++c) {
            double sum = 0;
            for (int t = 0; t < T; ++t) sum += ht[(i * T + t) * 3 + c];
            et += sum; es += hs[i * 3 + c]; diff += fabs(sum - hs[i * 3 + c]);
        }
    printf("paths %llu closest %llu shadow %llu contributions %llu | sum transient %.6f steady %.6f |diff| %.3e\n",
           (unsigned long long)cnt.paths, (unsigned long long)cnt.rays_closest, (unsigned long long)cnt.rays_shadow,
           (unsigned long long)cnt.splats_issued, et, es, diff);
    int ok = cnt.paths == (unsigned long long)W * H * SPP && cnt.rays_closest >= cnt.paths && cnt.splats_issued > 0 &&
             es > 0 && diff <= 1e-4 * es;
    /* ABI 9: the single-pass film lifecycle.  mtr_render_plan says which organisation runs and whether the row flush can store
     * DEVELOPED rows; then ONE mtr_render writes the (H,W,T,3) tensor — filled with garbage beforehand — with no clear and no
     * develop, and must give what clear + render + develop gave (f32 LDS sums: up to summation order) */
    uint32_t mode = 99u, dev_ok = 0u;
    CHECK(mtr_render_plan(scene, &p, &mode, &dev_ok));
    ok = ok && (mode == MTR_MODE_FUSED || mode == MTR_MODE_WAVEFRONT);
    if (dev_ok) {
        float *ht2 = (float *)malloc(nt * 12);
        if (hipMemset(t3, 0xff, nt * 12) || hipMemset(s4, 0, np * 16)) { fprintf(stderr, "memset failed\n"); return 1; }
        p.flags = MTR_FLAG_DEVELOPED_ROWS;
        CHECK(mtr_counters_reset(ctx));
        CHECK(mtr_render(scene, &p, t3, s4, &cnt, NULL));
        if (hipDeviceSynchronize() || hipMemcpy(ht2, t3, nt * 12, hipMemcpyDeviceToHost)) { fprintf(stderr, "copy back failed\n"); return 1; }
        double num = 0, den = 0;
        for (size_t i = 0; i < nt * 3; ++i) { const double dlt = (double)ht2[i] - ht[i]; num += dlt * dlt; den += (double)ht[i] * ht[i]; }
        printf("developed rows: relative L2 against clear + render + develop %.3e (mode %u)\n", sqrt(num / (den > 0 ? den : 1)), mode);
        ok = ok && num <= 1e-10 * den && cnt.paths == (unsigned long long)W * H * SPP;
        free(ht2);
    } else puts("developed rows: not offered for this plan");
    mtr_scene_destroy(scene); mtr_ctx_destroy(ctx);
    hipFree(t4); hipFree(s4); hipFree(t3); hipFree(s3); free(ht); free(hs);
    puts(ok ? "abi_smoke: PASS" : "abi_smoke: FAIL");
    return ok ? 0 : 2;
}
