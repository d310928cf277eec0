/*
 * client.c -- a plain C99 caller of libtriro_hip.so: no Python, no torch, no C++.
 *
 * What a maintainer of the reference would link against instead of the pybind11 module
 * (triro/backend/binding.cpp:31-59): device buffers from the HIP runtime, the C ABI of
 * include/triro_hip.h, and the reference's own test inputs (test/test.py:6-13, 47-64) with the
 * answers derived by hand in tests/golden/known_answers.json.  TEST-ONLY; built and run by
 * tests/test_c_client.py.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "triro_hip.h"

#define CHECK_HIP(x)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorName(e_)); \
            exit(2);                                                                      \
        }                                                                                 \
    } while (0)
#define CHECK_TR(x)                                                                       \
    do {                                                                                  \
        int s_ = (x);                                                                     \
        if (s_ != TR_OK) {                                                                \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, s_, tr_last_error()); \
            exit(3);                                                                      \
        }                                                                                 \
    } while (0)
#define EXPECT(c)                                                                         \
    do {                                                                                  \
        if (!(c)) {                                                                       \
            fprintf(stderr, "%s:%d: expectation failed: %s\n", __FILE__, __LINE__, #c);   \
            exit(4);                                                                      \
        }                                                                                 \
    } while (0)

static void *to_device(const void *h, size_t bytes) {
    void *d = NULL;
    CHECK_HIP(hipMalloc(&d, bytes ? bytes : 4));
    if (bytes) CHECK_HIP(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return d;
}
static void *device_zeros(size_t bytes) {
    void *d = NULL;
    CHECK_HIP(hipMalloc(&d, bytes ? bytes : 4));
    CHECK_HIP(hipMemset(d, 0xcd, bytes ? bytes : 4)); /* outputs must be fully written by the library */
    return d;
}
static void to_host(void *h, const void *d, size_t bytes) { CHECK_HIP(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); }

/* a flat [n,3] batch: shape / strides right-aligned as ray.cpp:151-159 fills RayInput */
static tr_rays flat_rays(const float *d_o, const float *d_d, int64_t n, int64_t ostride_ray) {
    tr_rays r;
    int k;
    memset(&r, 0, sizeof r);
    r.d_origins = d_o;
    r.d_directions = d_d;
    r.nray = n;
    for (k = 0; k < TR_MAX_SIZE_LENGTH; k++) { r.shape[k] = INT64_MAX; r.ostride[k] = 0; r.dstride[k] = 0; }
    r.shape[2] = n; r.shape[3] = 3;
    r.ostride[2] = ostride_ray; r.ostride[3] = 1;      /* ostride_ray = 0: one origin broadcast to every ray */
    r.dstride[2] = 3; r.dstride[3] = 1;
    return r;
}

static int feq(float a, float b) { return fabsf(a - b) <= 1e-6f; }

int main(void) {
    /* T2 of the survey (test/test.py:47-63): two parallel triangles at z = 0 and z = -1 */
    const float verts[18] = {0.5f, -0.5f, 0.f, 0.f, 0.5f, 0.f, -0.5f, -0.5f, 0.f,
                             0.5f, -0.5f, -1.f, 0.f, 0.5f, -1.f, -0.5f, -0.5f, -1.f};
    const int32_t faces[6] = {0, 1, 2, 3, 4, 5};
    const float orig[6] = {0.f, 0.f, -4.f, 0.f, 0.1f, 4.f};
    const float dirs[6] = {0.f, 0.f, 1.f, 0.f, 0.f, -1.f};
    float *d_v, *d_o, *d_d, *d_loc, *d_uv, *d_hloc;
    int32_t *d_f, *d_tri, *d_cnt, *d_hray, *d_htri;
    uint8_t *d_hit, *d_front;
    int64_t *d_off, *d_total, total = -1;
    tr_bvh *bvh = NULL;
    tr_bvh_info info;
    tr_rays rays;
    uint8_t hit[2], front[2];
    int32_t tri[2], cnt[2], hray[4], htri[4];
    float loc[6], uv[4], hloc[12];

    EXPECT(tr_abi_version() == TR_ABI_VERSION);
    CHECK_TR(tr_init(-1));
    d_v = (float *)to_device(verts, sizeof verts);
    d_f = (int32_t *)to_device(faces, sizeof faces);
    d_o = (float *)to_device(orig, sizeof orig);
    d_d = (float *)to_device(dirs, sizeof dirs);
    CHECK_TR(tr_bvh_build(d_v, 6, d_f, 2, NULL, &bvh));
    CHECK_TR(tr_bvh_get_info(bvh, &info));
    EXPECT(info.num_tris == 2 && info.num_nodes == 1);
    EXPECT(feq(info.aabb_min[2], -1.f) && feq(info.aabb_max[0], 0.5f));
    /* the handle keeps its own copy of the triangles (ALLOW_RANDOM_VERTEX_ACCESS, ray.cpp:37) */
    CHECK_HIP(hipMemset(d_v, 0, sizeof verts));
    CHECK_HIP(hipMemset(d_f, 0, sizeof faces));

    rays = flat_rays(d_o, d_d, 2, 3);
    d_hit = (uint8_t *)device_zeros(2); d_front = (uint8_t *)device_zeros(2);
    d_tri = (int32_t *)device_zeros(8); d_cnt = (int32_t *)device_zeros(8);
    d_loc = (float *)device_zeros(24); d_uv = (float *)device_zeros(16);

    CHECK_TR(tr_intersects_any(bvh, &rays, d_hit, NULL));
    to_host(hit, d_hit, 2);
    EXPECT(hit[0] == 1 && hit[1] == 1);
    CHECK_TR(tr_intersects_first(bvh, &rays, d_tri, NULL));
    to_host(tri, d_tri, 8);
    EXPECT(tri[0] == 1 && tri[1] == 0);
    CHECK_TR(tr_intersects_closest(bvh, &rays, d_hit, d_front, d_tri, d_loc, d_uv, NULL));
    to_host(hit, d_hit, 2); to_host(front, d_front, 2); to_host(tri, d_tri, 8); to_host(loc, d_loc, 24); to_host(uv, d_uv, 16);
    EXPECT(hit[0] == 1 && front[0] == 0 && tri[0] == 1);      /* from below: back face of the lower triangle */
    EXPECT(feq(loc[0], 0.f) && feq(loc[1], 0.f) && feq(loc[2], -1.f) && feq(uv[0], 0.25f) && feq(uv[1], 0.5f));
    EXPECT(hit[1] == 1 && front[1] == 1 && tri[1] == 0);
    EXPECT(feq(loc[3], 0.f) && feq(loc[4], 0.1f) && feq(loc[5], 0.f) && feq(uv[2], 0.2f) && feq(uv[3], 0.6f));
    CHECK_TR(tr_intersects_count(bvh, &rays, d_cnt, NULL));
    to_host(cnt, d_cnt, 8);
    EXPECT(cnt[0] == 2 && cnt[1] == 2);

    /* intersectsLocation: count -> clamp / scan -> fill (ray.cpp:324-378) */
    d_off = (int64_t *)device_zeros(16); d_total = (int64_t *)device_zeros(8);
    CHECK_TR(tr_hits_scan(d_cnt, 2, TR_MAX_ANYHIT_SIZE, d_off, d_total, &total, NULL));
    EXPECT(total == 4);
    d_hloc = (float *)device_zeros(48); d_hray = (int32_t *)device_zeros(16); d_htri = (int32_t *)device_zeros(16);
    CHECK_TR(tr_intersects_location_fill(bvh, &rays, TR_MAX_ANYHIT_SIZE, d_off, d_hloc, d_hray, d_htri, 0, NULL));
    to_host(hloc, d_hloc, 48); to_host(hray, d_hray, 16); to_host(htri, d_htri, 16);
    EXPECT(hray[0] == 0 && hray[1] == 0 && hray[2] == 1 && hray[3] == 1);
    EXPECT(htri[0] == 1 && htri[1] == 0 && htri[2] == 0 && htri[3] == 1);      /* nearest first */
    EXPECT(feq(hloc[2], -1.f) && feq(hloc[5], 0.f) && feq(hloc[8], 0.f) && feq(hloc[11], -1.f));

    /* closest hit as 12-byte records and back (ABI 6 face form, ABI 7 slot form): what a ray-sharded run exchanges */
    {
        tr_packed_hit *d_rec = (tr_packed_hit *)device_zeros(2 * sizeof(tr_packed_hit)), rec[2];
        float *d_v2 = (float *)to_device(verts, sizeof verts);        /* (the mesh arrays were zeroed above) */
        int32_t *d_f2 = (int32_t *)to_device(faces, sizeof faces);
        int form;
        for (form = 0; form < 3; form++) {
            CHECK_HIP(hipMemset(d_hit, 7, 2)); CHECK_HIP(hipMemset(d_loc, 0xff, 24));
            if (form == 0) {
                CHECK_TR(tr_intersects_closest_packed(bvh, &rays, d_rec, NULL));
                CHECK_TR(tr_closest_expand(d_rec, 2, d_v2, 6, d_f2, 2, d_hit, d_front, d_tri, d_loc, d_uv, NULL));
            } else {
                CHECK_TR(tr_intersects_closest_packed_slots(bvh, &rays, d_rec, NULL));
                if (form == 1) CHECK_TR(tr_closest_expand_slots(bvh, d_rec, 2, d_hit, d_front, d_tri, d_loc, d_uv, NULL));
                else CHECK_TR(tr_closest_expand_slots_rows(bvh, d_rec, 2, 2, d_hit, d_front, d_tri, d_loc, d_uv, NULL));
            }
            to_host(rec, d_rec, sizeof rec);
            EXPECT((rec[0].tri & 0x80000000u) == 0 && ((rec[0].tri >> 30) & 1u) == 0 && ((rec[1].tri >> 30) & 1u) == 1);
            to_host(hit, d_hit, 2); to_host(front, d_front, 2); to_host(tri, d_tri, 8); to_host(loc, d_loc, 24); to_host(uv, d_uv, 16);
            EXPECT(hit[0] == 1 && front[0] == 0 && tri[0] == 1 && hit[1] == 1 && front[1] == 1 && tri[1] == 0);
            EXPECT(feq(loc[2], -1.f) && feq(uv[0], 0.25f) && feq(uv[1], 0.5f) && feq(loc[4], 0.1f) && feq(uv[2], 0.2f) && feq(uv[3], 0.6f));
        }
        CHECK_HIP(hipFree(d_rec)); CHECK_HIP(hipFree(d_v2)); CHECK_HIP(hipFree(d_f2));
    }
    /* ... and as 4-byte records for a destination that holds the rays (ABI 8): the slot, then the end of the query */
    {
        int32_t *d_slot = (int32_t *)device_zeros(8), slot[2];
        CHECK_HIP(hipMemset(d_hit, 7, 2)); CHECK_HIP(hipMemset(d_loc, 0xff, 24));
        CHECK_TR(tr_intersects_closest_slots(bvh, &rays, d_slot, NULL));
        to_host(slot, d_slot, 8);
        EXPECT(slot[0] >= 0 && slot[0] < 2 && slot[1] >= 0 && slot[1] < 2 && slot[0] != slot[1]);
        CHECK_TR(tr_closest_from_slots(bvh, &rays, d_slot, 0, d_hit, d_front, d_tri, d_loc, d_uv, NULL));
        to_host(hit, d_hit, 2); to_host(front, d_front, 2); to_host(tri, d_tri, 8); to_host(loc, d_loc, 24); to_host(uv, d_uv, 16);
        EXPECT(hit[0] == 1 && front[0] == 0 && tri[0] == 1 && hit[1] == 1 && front[1] == 1 && tri[1] == 0);
        EXPECT(feq(loc[2], -1.f) && feq(uv[0], 0.25f) && feq(uv[1], 0.5f) && feq(loc[4], 0.1f) && feq(uv[2], 0.2f) && feq(uv[3], 0.6f));
        EXPECT(tr_closest_from_slots(bvh, &rays, NULL, 0, d_hit, d_front, d_tri, d_loc, d_uv, NULL) == TR_ERR_INVALID_ARG);
        CHECK_HIP(hipFree(d_slot));
    }

    /* T1 (test/test.py:6-13) through update: one triangle, a hit and a miss; broadcast origin on the second call */
    {
        const float v1[9] = {0.5f, -0.5f, 0.f, 0.f, 0.5f, 0.f, -0.5f, -0.5f, 0.f};
        const int32_t f1[3] = {0, 1, 2};
        const float o1[6] = {0.f, 0.f, 4.f, 10.f, 10.f, 10.f};
        const float dd1[6] = {0.f, 0.f, -1.f, 0.f, 1.f, 0.f};
        CHECK_HIP(hipMemcpy(d_v, v1, sizeof v1, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_f, f1, sizeof f1, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_o, o1, sizeof o1, hipMemcpyHostToDevice));
        CHECK_HIP(hipMemcpy(d_d, dd1, sizeof dd1, hipMemcpyHostToDevice));
        CHECK_TR(tr_bvh_update(bvh, d_v, 3, d_f, 1, NULL));
        CHECK_TR(tr_intersects_closest(bvh, &rays, d_hit, d_front, d_tri, d_loc, d_uv, NULL));
        to_host(hit, d_hit, 2); to_host(front, d_front, 2); to_host(tri, d_tri, 8); to_host(loc, d_loc, 24); to_host(uv, d_uv, 16);
        EXPECT(hit[0] == 1 && front[0] == 1 && tri[0] == 0 && feq(loc[2], 0.f) && feq(uv[0], 0.25f) && feq(uv[1], 0.5f));
        EXPECT(hit[1] == 0 && front[1] == 0 && tri[1] == -1 && loc[3] == 0.f && loc[4] == 0.f && loc[5] == 0.f && uv[2] == 0.f && uv[3] == 0.f);
        rays = flat_rays(d_o, d_d, 2, 0);           /* origin (0,0,4) for both rays: stride 0 (README.md:35-39) */
        CHECK_TR(tr_intersects_any(bvh, &rays, d_hit, NULL));
        to_host(hit, d_hit, 2);
        EXPECT(hit[0] == 1 && hit[1] == 0);         /* (0,0,4) + t (0,1,0) passes above the triangle */
    }

    /* errors are status codes with a message, never exit() (the reference: optix8.h:41-60) */
    EXPECT(tr_intersects_any(NULL, &rays, d_hit, NULL) == TR_ERR_INVALID_ARG);
    EXPECT(strlen(tr_last_error()) > 0);
    EXPECT(tr_set_option("no_such_option", 1) == TR_ERR_INVALID_ARG);
    {
        const int32_t bad[3] = {0, 1, 7};            /* vertex index out of range */
        tr_bvh *b2 = NULL;
        CHECK_HIP(hipMemcpy(d_f, bad, sizeof bad, hipMemcpyHostToDevice));
        EXPECT(tr_bvh_build(d_v, 3, d_f, 1, NULL, &b2) == TR_ERR_INVALID_ARG && b2 == NULL);
    }
    CHECK_TR(tr_bvh_destroy(bvh));
    CHECK_HIP(hipDeviceSynchronize());
    printf("C CLIENT OK\n");
    return 0;
}
