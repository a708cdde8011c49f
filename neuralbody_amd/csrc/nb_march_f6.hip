// nb_march_f6.hip — the "f16f6" march kernel: nb_march_f16.hip compiled with the cross terms in six bits (fp6 e2m3 weights x
// bf6 e3m2 activations, block scales), see the header of that file.
#define F_SIX 1
#include "nb_march_f16.hip"
