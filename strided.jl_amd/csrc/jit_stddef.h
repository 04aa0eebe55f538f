// <stddef.h> as seen by hiprtc (which ships <stdint.h>/<cstdint> but not this one): the embedded
// copy of include/strided_hip.h only needs size_t.
#pragma once
typedef __SIZE_TYPE__ size_t;
