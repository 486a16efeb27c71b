// The second translation unit of fused_voice.hip: the row walkers of Square, Sawtooth and Triangle (see the note at the top of
// that file); compiled in parallel with the first, which holds the Sine kernels and the C ABI.
#define SIG_FUSED_PART_B 1
#include "fused_voice.hip"
