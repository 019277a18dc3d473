/*
 * wga_kernels2.h — the consumers beside paf2maf / stat, one header per kernel family (each names the reference code it
 * replaces); included in dependency order: a part may use helpers of the parts in front of it.
 */
#ifndef WGA_KERNELS2_H
#define WGA_KERNELS2_H

#include "wga_kernels.h"
#include <type_traits>

#include "wga_k_class.h"
#include "wga_k5_pafcov.h"
#include "wga_k6_pafpseudo.h"
#include "wga_k3_maf.h"
#include "wga_k7_paf_call.h"
#include "wga_k8_tokenise.h"
#include "wga_k9_bed.h"
#include "wga_k10_chain.h"
#include "wga_k11_bridges.h"
#include "wga_k12_dotplot.h"
#include "wga_k13_splitters.h"
#include "wga_k15_fasta.h"
#include "wga_k18_bgzf_deflate.h"

#endif /* WGA_KERNELS2_H */
