// libpylda_hip.so - estep_quad.h without the hand-over to the live-topic kernel (template flag HANDOFF = false): the
// instantiations of launch_quad.hip again, as a translation unit of its own so that the two compile side by side.
// (host side of the C ABI declared in include/pylda_hip.h; see host_internal.h for the map of the translation units)
#define PYLDA_QUAD_HANDOFF false
#define PYLDA_QUAD_LAUNCHER launch_quad_dense_any
#include "launch_quad.hip"
