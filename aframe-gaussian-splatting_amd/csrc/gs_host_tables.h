// gs_host_tables.h -- host-built constant tables uploaded to the device at gs_create().
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include "gs_device_math.h"

#define GS_POW10_ENTRIES ((GS_POW10_EMAX - GS_POW10_EMIN + 1) * 9)

// tab[(E-6)*9 + (D-1)] = the double nearest to D * 10^-E (correctly rounded by strtod); see
// gsm::js_parse_int.  Entries that underflow to 0 are left 0 (unreachable for pack inputs).
static inline void gs_build_pow10_table(double *tab)
{
    char buf[32];
    for (int E = GS_POW10_EMIN; E <= GS_POW10_EMAX; E++)
        for (int D = 1; D <= 9; D++) {
            snprintf(buf, sizeof buf, "%de-%d", D, E);
            tab[(E - GS_POW10_EMIN) * 9 + (D - 1)] = strtod(buf, NULL);
        }
}
