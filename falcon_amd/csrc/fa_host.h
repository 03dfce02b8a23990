// fa_host.h -- host-only helpers shared between the library's translation units.
#pragma once
#include <string>

// FASTA records of one consensus string (fasta.cpp; modes: FA_FASTA_* of falcon_amd.h)
void fa_fasta_append(std::string &out, const char *seed_id, const char *cns, long long n, int mode);

// ASCII -> 2 bits per base, 16 bases per u32 (pack_host.cpp): out[0 .. n_out) = the packed
// sequence and zero words behind it; returns the position of the first byte other than
// upper-case A, C, G, T, or -1
int fa_pack_host(const char *s, int len, unsigned *out, long long n_out);
