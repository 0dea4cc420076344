/* A plain C consumer of include/pbhip.h + libpbhip.so (no Python, no torch): what a cgo / JNI / FFI binding links.
 * Creates a plan, generates a chirp, dedisperses a host block of ones and prints the status of each call and one
 * output sample.  Built and run by tests/test_abi.py. */
#include <stdio.h>
#include <stdlib.h>
#include "pbhip.h"

int main(void) {
    const int64_t n = 1 << 16;
    const int nchan = 2, npol = 2;
    printf("version: %s\n", pbh_version());
    printf("devices: %d\n", pbh_device_count());
    pbh_plan* plan = NULL;
    int rc = pbh_plan_create(&plan, 0, n, nchan, npol, PBH_C64, 100, n - 100);
    printf("plan_create: %d\n", rc);
    if (rc != PBH_OK) {
        printf("error: %s\n", pbh_last_error());
        return 0;
    }
    const double freqs[2] = {0.9995e9, 1.0005e9};
    rc = pbh_chirp_generate(plan, 1.0 / 2.41e-4 * 1e12, 1e-6, freqs, 1e9);
    printf("chirp_generate: %d\n", rc);
    float* in = (float*)malloc(sizeof(float) * 2 * n * nchan * npol);
    float* out = (float*)malloc(sizeof(float) * 2 * (n - 200) * nchan * npol);
    for (int64_t i = 0; i < 2 * n * nchan * npol; ++i) in[i] = (i & 1) ? 0.f : 1.f;   /* every sample = 1 + 0i */
    rc = pbh_dedisperse(plan, in, out, PBH_HOST, PBH_HOST);
    printf("dedisperse: %d\n", rc);
    /* a constant series is the DC bin: its chirp phase is exp(-i phi(f_chan)), a unit-magnitude factor */
    printf("abs2 of one output sample: %.6f\n", (double)(out[0] * out[0] + out[1] * out[1]));
    pbh_plan_destroy(plan);
    free(in);
    free(out);
    return 0;
}
