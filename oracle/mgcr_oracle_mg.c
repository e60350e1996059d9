/* TEST INFRASTRUCTURE — CPU oracle (multigrid part), NOT part of the product path. */
#include <complex.h>
#include <stdio.h>
#include <stdlib.h>
typedef double _Complex cplx;
typedef struct orc_mg orc_mg;
void orc_mg_apply(orc_mg *mg, const cplx *f, cplx *y) {
    (void)mg; (void)f; (void)y;
    fprintf(stderr, "orc_mg_apply: multigrid oracle not built yet\n");
    abort();
}
