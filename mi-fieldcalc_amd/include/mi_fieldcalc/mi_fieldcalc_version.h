// Version of the operator API this library is source-compatible with
// (reference: src/mi_fieldcalc/mi_fieldcalc_version.h:33-40, release 0.1.9).
#ifndef MI_FIELDCALC_VERSION_H
#define MI_FIELDCALC_VERSION_H

#define MI_FIELDCALC_VERSION_MAJOR 0
#define MI_FIELDCALC_VERSION_MINOR 1
#define MI_FIELDCALC_VERSION_PATCH 9

#define MI_FIELDCALC_VERSION_INT(major, minor, patch) (1000000 * major + 1000 * minor + patch)
#define MI_FIELDCALC_VERSION_CURRENT_INT \
  MI_FIELDCALC_VERSION_INT(MI_FIELDCALC_VERSION_MAJOR, MI_FIELDCALC_VERSION_MINOR, MI_FIELDCALC_VERSION_PATCH)

// set by the gfx950 implementation only
#define MI_FIELDCALC_BACKEND_HIP_GFX950 1

#endif // MI_FIELDCALC_VERSION_H
