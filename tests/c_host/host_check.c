/* A plain C host of libkanter_core_amd.so -- no Python, no torch: what a Rust `extern "C"` binding of the
 * reference's per-node functions would do, checked against the CPU oracle's C functions.
 * Test infrastructure (it links the oracle); built and run by tests/test_gpu_c_host.py:
 *   gcc -std=c11 -O1 host_check.c -I include -L kanter_core_amd -lkanter_core_amd -L oracle -lkc_oracle -lm */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kanter_core_amd.h"

/* oracle/kc_oracle.c */
int kco_mix_plane(int op, const float *l, const float *r, float *out, size_t n);
void kco_rgba_to_gray(const float *r, const float *g, const float *b, float *out, size_t n);
int kco_resize_plane(const float *src, uint32_t sw, uint32_t sh, float *dst, uint32_t dw, uint32_t dh, int filter);
void kco_height_to_normal(const float *hgt, uint32_t w, uint32_t h, float *nx, float *ny, float *nz);
void kco_to_u8_rgba(const float *r, const float *g, const float *b, const float *a, size_t n, int srgb, uint8_t *out);

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int s_ = (call);                                                                              \
        if (s_ != KC_OK) {                                                                            \
            fprintf(stderr, "%s:%d %s -> %d %s (%s)\n", __FILE__, __LINE__, #call, s_, kc_status_string(s_), kc_last_error()); \
            exit(1);                                                                                  \
        }                                                                                             \
    } while (0)

static uint64_t rng_state = 0x5EED0001u;
static float rnd(void)
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((rng_state >> 40) * (1.0 / 16777216.0)) * 1.5f - 0.25f;
}

static float *plane(size_t n)
{
    float *p = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) p[i] = rnd();
    return p;
}

static int same(const float *a, const float *b, size_t n, const char *what)
{
    if (memcmp(a, b, n * sizeof(float)) == 0) return 0;
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += memcmp(a + i, b + i, 4) != 0;
    fprintf(stderr, "MISMATCH %s: %zu of %zu floats differ\n", what, bad, n);
    return 1;
}

static void download(kc_image *img, int n_planes, float **out, size_t n)
{
    for (int i = 0; i < n_planes; ++i) out[i] = (float *)malloc(n * sizeof(float));
    CHECK(kc_image_to_f32(img, out, n_planes));
}

int main(void)
{
    enum { W = 200, H = 120, SW = 50, SH = 30 };
    const size_t n = (size_t)W * H, sn = (size_t)SW * SH;
    int fails = 0;
    CHECK(kc_init(0));

    float *a[4], *b[4], *small[1];
    for (int c = 0; c < 4; ++c) a[c] = plane(n), b[c] = plane(n);
    small[0] = plane(sn);
    kc_image *A, *B, *S;
    CHECK(kc_image_from_f32((const float *const *)a, 4, W, H, &A));
    CHECK(kc_image_from_f32((const float *const *)b, 4, W, H, &B));
    CHECK(kc_image_from_f32((const float *const *)small, 1, SW, SH, &S));

    /* mix::process, src/node/mix.rs:51: ((A + B) * A) / B on R, G, B; alpha = 1 */
    kc_image *m1, *m2, *m3;
    CHECK(kc_mix_process(A, B, KC_MIX_ADD, &m1));
    CHECK(kc_mix_process(m1, A, KC_MIX_MULTIPLY, &m2));
    CHECK(kc_mix_process(m2, B, KC_MIX_DIVIDE, &m3));
    float *got[4], *t1 = malloc(n * 4), *t2 = malloc(n * 4), *want = malloc(n * 4);
    download(m3, 4, got, n);
    for (int c = 0; c < 3; ++c) {
        kco_mix_plane(0, a[c], b[c], t1, n);
        kco_mix_plane(2, t1, a[c], t2, n);
        kco_mix_plane(3, t2, b[c], want, n);
        fails += same(got[c], want, n, "mix chain");
    }
    for (size_t i = 0; i < n; ++i) fails += got[3][i] != 1.0f;

    /* resize_buffers + as_type + Mix(Subtract): gray 50x30 -> 200x120 (Triangle), RGBA minus gray */
    kc_image *up, *sub;
    kc_size size = { W, H };
    CHECK(kc_resize_image(S, size, KC_FILTER_TRIANGLE, &up));
    CHECK(kc_mix_process(A, up, KC_MIX_SUBTRACT, &sub));
    float *gs[4], *upw = malloc(n * 4);
    download(sub, 4, gs, n);
    kco_resize_plane(small[0], SW, SH, upw, W, H, 1);
    for (int c = 0; c < 3; ++c) {
        kco_mix_plane(1, a[c], upw, want, n);
        fails += same(gs[c], want, n, "resize + subtract");
    }

    /* SeparateRgba -> HeightToNormal on the red plane -> to_u8 */
    kc_image *ch[4], *nrm;
    CHECK(kc_separate_rgba_process(A, ch));
    CHECK(kc_height_to_normal_process(ch[0], &nrm));
    float *gn[4], *nx = malloc(n * 4), *ny = malloc(n * 4), *nz = malloc(n * 4);
    download(nrm, 4, gn, n);
    kco_height_to_normal(a[0], W, H, nx, ny, nz);
    fails += same(gn[0], nx, n, "normal x") + same(gn[1], ny, n, "normal y") + same(gn[2], nz, n, "normal z");
    uint8_t *u8 = malloc(n * 4), *u8w = malloc(n * 4);
    CHECK(kc_image_to_u8(nrm, 0, u8));
    kco_to_u8_rgba(nx, ny, nz, gn[3], n, 0, u8w);
    if (memcmp(u8, u8w, n * 4) != 0) fails += 1, fprintf(stderr, "MISMATCH to_u8\n");

    /* the graph API: Embed(A), Embed(B) -> Mix(Multiply) -> Mix(Subtract)(Value 1, x) -> OutputRgba */
    kc_tex_pro *tp;
    kc_live_graph *lg;
    CHECK(kc_tex_pro_new(10000000, &tp));
    CHECK(kc_tex_pro_new_live_graph(tp, &lg));
    CHECK(kc_live_graph_embed_slot_data_with_id(lg, A, 0, 0));
    CHECK(kc_live_graph_embed_slot_data_with_id(lg, B, 0, 1));
    kc_node_desc d;
    uint32_t na, nb, nm, nv, ni, no;
    memset(&d, 0, sizeof d);
    d.resize_policy = KC_POLICY_MOST_PIXELS, d.resize_filter = KC_FILTER_TRIANGLE;
    d.node_type = KC_NODE_EMBED, d.embed_id = 0;
    CHECK(kc_live_graph_add_node(lg, &d, &na));
    d.embed_id = 1;
    CHECK(kc_live_graph_add_node(lg, &d, &nb));
    d.node_type = KC_NODE_MIX, d.mix_type = KC_MIX_MULTIPLY;
    CHECK(kc_live_graph_add_node(lg, &d, &nm));
    d.node_type = KC_NODE_VALUE, d.value = 1.0f;
    CHECK(kc_live_graph_add_node(lg, &d, &nv));
    d.node_type = KC_NODE_MIX, d.mix_type = KC_MIX_SUBTRACT;
    CHECK(kc_live_graph_add_node(lg, &d, &ni));
    d.node_type = KC_NODE_OUTPUT_RGBA, d.text = "out";
    CHECK(kc_live_graph_add_node(lg, &d, &no));
    CHECK(kc_live_graph_connect(lg, na, nm, 0, 0));
    CHECK(kc_live_graph_connect(lg, nb, nm, 0, 1));
    CHECK(kc_live_graph_connect(lg, nv, ni, 0, 0));
    CHECK(kc_live_graph_connect(lg, nm, ni, 0, 1));
    CHECK(kc_live_graph_connect(lg, ni, no, 0, 0));
    CHECK(kc_live_graph_await_clean(lg, no));
    kc_image *res;
    CHECK(kc_live_graph_slot_data(lg, no, 0, &res));
    int is_rgba = -1;
    CHECK(kc_image_is_rgba(res, &is_rgba));
    /* the gray Value(1) on the left makes the result gray: right = ((r + g) + b) / 3 of A * B (mix.rs:57-62) */
    float *gg[1], *prod[3], *gray = malloc(n * 4), *ones = malloc(n * 4);
    download(res, 1, gg, n);
    for (int c = 0; c < 3; ++c) {
        prod[c] = malloc(n * 4);
        kco_mix_plane(2, a[c], b[c], prod[c], n);
    }
    kco_rgba_to_gray(prod[0], prod[1], prod[2], gray, n);
    for (size_t i = 0; i < n; ++i) ones[i] = 1.0f;
    kco_mix_plane(1, ones, gray, want, n);
    fails += (is_rgba != 0) + same(gg[0], want, n, "graph: invert of a product");
    /* round 2: the multi-GPU plan of this graph (host only), row bands of its result, the specialiser's knobs -- from C */
    {
        kc_partition *plan;
        CHECK(kc_live_graph_partition(lg, no, 4, KC_PARTITION_SPREAD, &plan));
        int world = 0, home = -1, levels = 0;
        CHECK(kc_partition_info(plan, &world, &home, &levels));
        uint32_t n_nodes = 0, n_xfer = 0;
        CHECK(kc_partition_nodes(plan, NULL, 0, &n_nodes));
        CHECK(kc_partition_transfers(plan, NULL, 0, &n_xfer));
        kc_placement pl[16];
        CHECK(kc_partition_nodes(plan, pl, 16, &n_nodes));
        int replicated = 0, sources = 0;
        for (uint32_t i = 0; i < n_nodes; ++i) {
            replicated += pl[i].kind == KC_KIND_REPLICATED && pl[i].rank == -1;
            sources += pl[i].kind == KC_KIND_SOURCE;
        }
        /* one chain, nothing to cut: 6 nodes, the Value replicated, both embeds sources, no transfer */
        if (world != 4 || home != 0 || n_nodes != 6 || n_xfer != 0 || replicated != 1 || sources != 2) {
            fails += 1;
            fprintf(stderr, "partition: world %d home %d nodes %u transfers %u replicated %d sources %d\n", world, home, n_nodes, n_xfer, replicated, sources);
        }
        CHECK(kc_partition_free(plan));

        kc_band_rows need[4];
        uint32_t n_need = 0;
        CHECK(kc_live_graph_band_source_rows(lg, no, 40, 80, need, 4, &n_need));
        for (uint32_t i = 0; i < n_need; ++i)
            if (need[i].y0 != 40 || need[i].y1 != 80 || need[i].width != W || need[i].height != H) fails += 1, fprintf(stderr, "band rows of node %u\n", need[i].node_id);
        if (n_need != 2) fails += 1, fprintf(stderr, "expected the two embedded sources, got %u\n", n_need);
        kc_image *band;
        CHECK(kc_live_graph_evaluate_band(lg, no, 0, 40, 80, &band));
        kc_size bs;
        CHECK(kc_image_size(band, &bs));
        float *gb[1];
        download(band, 1, gb, (size_t)W * 40);
        fails += (bs.width != W || bs.height != 40) + same(gb[0], gg[0] + (size_t)40 * W, (size_t)W * 40, "rows 40..80 of the graph's result");
        CHECK(kc_image_release(band));

        CHECK(kc_set_specialize(2, 0)); /* compile at first sight: the next evaluation of this chain uses the generated kernel */
        CHECK(kc_live_graph_connect(lg, na, nm, 0, 0));
        CHECK(kc_live_graph_await_clean(lg, no));
        kc_image *res2;
        CHECK(kc_live_graph_slot_data(lg, no, 0, &res2));
        float *g2[1];
        download(res2, 1, g2, n);
        uint64_t compiled = 0, failed = 0, spec_launches = 0, pending = 0;
        CHECK(kc_specialize_stats(&compiled, &failed, &spec_launches, &pending));
        fails += same(g2[0], want, n, "specialised kernel") + (failed != 0) + (spec_launches == 0);
        CHECK(kc_image_release(res2));
        CHECK(kc_set_specialize(1, 2));
    }

    /* an error path: an unknown node id */
    if (kc_live_graph_await_clean(lg, 12345) != KC_ERR_INVALID_NODE_ID) fails += 1, fprintf(stderr, "expected InvalidNodeId\n");

    uint64_t in_use, cached, launches;
    CHECK(kc_stats(&in_use, &cached, &launches));
    kc_image *all[] = { A, B, S, m1, m2, m3, up, sub, ch[0], ch[1], ch[2], ch[3], nrm, res };
    for (size_t i = 0; i < sizeof all / sizeof all[0]; ++i) CHECK(kc_image_release(all[i]));
    CHECK(kc_live_graph_free(lg));
    CHECK(kc_tex_pro_free(tp));
    CHECK(kc_stats(&in_use, &cached, NULL));
    if (in_use != 0) fails += 1, fprintf(stderr, "leak: %llu bytes still in use\n", (unsigned long long)in_use);
    CHECK(kc_shutdown());
    printf("c host: %d failures, %llu kernel launches\n", fails, (unsigned long long)launches);
    return fails != 0;
}
