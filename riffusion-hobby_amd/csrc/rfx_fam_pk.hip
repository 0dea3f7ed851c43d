// rfx_fam_pk.hip - translation unit 1 of the row-family kernels: the Griffin-Lim kernels of every geometry but 48 kHz, compiled
// with packed fp32 butterflies (v_pk_*_f32 on (re, im) register pairs, rfx_core.h).  See the head of rfx_fam.hip for why the
// file is compiled twice and what each unit holds.
#define RFX_FAM_TU 1
#ifndef RFX_NO_PK
#define RFX_PK 1
#endif
#include "rfx_fam.hip"
