// Undefined-value vocabulary of the operator API.
// Source-compatible with the reference's src/mi_fieldcalc/FieldDefined.h:35-47.
#ifndef MI_FIELDCALC_FIELDDEFINED_H
#define MI_FIELDCALC_FIELDDEFINED_H

#include <cstdlib>

extern const float fieldUndef; // == miutil::UNDEF

namespace miutil {

extern const float UNDEF; // 1.0e35f

// State of a field with respect to the undefined value.  Operators take it
// IN (state of the inputs; ALL_DEFINED switches the per-cell tests off) and
// give it back OUT (state of the result).
enum ValuesDefined { ALL_DEFINED = 0, NONE_DEFINED, SOME_DEFINED };

// scan of a host array: a cell counts as defined when it is < UNDEF
ValuesDefined checkDefined(const float* data, size_t n);
// from a count of undefined cells out of n
ValuesDefined checkDefined(size_t n_undefined, size_t n);
// state of a result computed from two inputs in states a and b
ValuesDefined combineDefined(ValuesDefined a, ValuesDefined b);

} // namespace miutil

#endif // MI_FIELDCALC_FIELDDEFINED_H
