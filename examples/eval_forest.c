/* A plain-C consumer of librdf_hip.so: what the drop-in boundary looks like without Python.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/eval_forest.c \
 *       -L 3d-beats_amd/csrc -l:librdf_hip.so -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/3d-beats_amd/csrc -Wl,-rpath,/opt/rocm/lib -o eval_forest_example && ./eval_forest_example
 *
 * One 8x6 depth frame, one tree of depth 1 (the root, both sides leaves).  The root compares the pixel with itself
 * (u = v = 0), so f = 0 < thresh = 1 sends every valid pixel left, whose PDF (0.2, 0.8) makes label 1; pixels with
 * depth 0 or 65535 keep the caller's pre-fill (tree_eval.cu:88-89). */
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "rdf_hip.h"

#define W 8
#define H 6

int main(void)
{
    uint16_t depth[H * W], labels[H * W];
    float forest[1 * 1 * (7 + 2 * 2)] = {0.f, 0.f, 0.f, 0.f, /* thresh */ 1.f, /* l_next, r_next: leaves */ 0.f, 0.f,
                                         /* left pdf */ 0.2f, 0.8f, /* right pdf */ 0.9f, 0.1f};
    void *d_depth = NULL, *d_labels = NULL, *d_forest = NULL;
    int i, rc, bad = 0;

    if (rdf_abi_version() < 1) return 2;
    for (i = 0; i < H * W; ++i) depth[i] = (uint16_t)(3000 + i);
    depth[5] = 0;
    depth[17] = 65535;
    memset(labels, 0xFF, sizeof labels);               /* the caller's pre-fill: 65535 */

    if (hipMalloc(&d_depth, sizeof depth) != hipSuccess || hipMalloc(&d_labels, sizeof labels) != hipSuccess ||
        hipMalloc(&d_forest, sizeof forest) != hipSuccess) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    hipMemcpy(d_depth, depth, sizeof depth, hipMemcpyHostToDevice);
    hipMemcpy(d_labels, labels, sizeof labels, hipMemcpyHostToDevice);
    hipMemcpy(d_forest, forest, sizeof forest, hipMemcpyHostToDevice);

    rc = rdf_eval_forest((const uint16_t *)d_depth, 1, W, H, (const float *)d_forest, 1, 1, 2, NULL, -1,
                         (uint16_t *)d_labels, 1, 1.0f, NULL);
    if (rc != RDF_OK) {
        fprintf(stderr, "rdf_eval_forest: %s\n", rdf_error_string(rc));
        return 1;
    }
    hipDeviceSynchronize();
    hipMemcpy(labels, d_labels, sizeof labels, hipMemcpyDeviceToHost);
    for (i = 0; i < H * W; ++i) {
        const unsigned want = (i == 5 || i == 17) ? 65535u : 1u;
        if (labels[i] != want) ++bad;
    }
    /* The life of a packed table (include/rdf_hip.h, "Load-time repack"): allocate rdf_forest_packed_bytes(), pack once at model
     * load, evaluate from it as often as needed, and TELL the library before the memory goes away -- it remembers what it read
     * from a table's info block per (device, address), and an allocation that later lands on the same address must not meet
     * that memory (a consumer that forgets is told: the call after such a launch returns RDF_ERR_STALE). */
    {
        void *d_packed = NULL;
        const size_t nbytes = rdf_forest_packed_bytes(1, 1, 2);
        float scale = 0.f;
        int exact = -1;
        if (nbytes == 0 || hipMalloc(&d_packed, nbytes) != hipSuccess) ++bad;       /* (hipMalloc is 256-byte aligned: >= the 128 needed) */
        else {
            hipMemcpy(d_labels, memset(labels, 0xFF, sizeof labels), sizeof labels, hipMemcpyHostToDevice);
            rc = rdf_forest_pack((const float *)d_forest, 1, 1, 2, 1.0f, d_packed, NULL);
            if (rc == RDF_OK) rc = rdf_forest_info(d_packed, 1, 1, 2, NULL, NULL, &exact, &scale);
            if (rc == RDF_OK)
                rc = rdf_eval_forest_packed((const uint16_t *)d_depth, 1, W, H, d_packed, (const float *)d_forest, 1, 1, 2, NULL, -1,
                                            (uint16_t *)d_labels, 1, NULL);
            if (rc != RDF_OK) {
                fprintf(stderr, "packed path: %s\n", rdf_error_string(rc));
                ++bad;
            }
            hipDeviceSynchronize();
            hipMemcpy(labels, d_labels, sizeof labels, hipMemcpyDeviceToHost);
            for (i = 0; i < H * W; ++i)
                if (labels[i] != ((i == 5 || i == 17) ? 65535u : 1u)) ++bad;
            if (scale != 1.0f || exact != 0) ++bad;
            if (rdf_forest_forget(d_packed) != RDF_OK) ++bad;      /* before the free */
            hipFree(d_packed);
        }
    }
    /* argument errors come back as negative codes, not as faults */
    if (rdf_eval_forest(NULL, 1, W, H, (const float *)d_forest, 1, 1, 2, NULL, -1, (uint16_t *)d_labels, 1, 1.0f, NULL) !=
        RDF_ERR_NULL_PTR)
        ++bad;
    hipFree(d_depth); hipFree(d_labels); hipFree(d_forest);
    printf(bad ? "FAIL: %d mismatches\n" : "PASS\n", bad);
    return bad ? 1 : 0;
}
