// fa_host.h -- host-only helpers shared between the library's translation units.
#pragma once
#include <string>

// FASTA records of one consensus string (fasta.cpp; modes: FA_FASTA_* of falcon_amd.h)
void fa_fasta_append(std::string &out, const char *seed_id, const char *cns, long long n, int mode);
