// lutcheck.cpp -- pins the oracle's 2-bit decode against the REFERENCE'S OWN generated tables.
// Built only when /root/reference is present (oracle/Makefile: lutcheck); includes the reference
// headers where they lie (dotp_lut.hpp:3,1030,2057 ; na_lut.hpp:3), copies nothing.
// Output binary goes to oracle/_ref/ (git-ignored).  Exit code 0 = all 4160 entries identical.
#include <cstdio>
#include "dotp_lut.hpp"
#include "na_lut.hpp"
#include "gv_oracle.hpp"

int main() {
    long bad = 0, n = 0;
    for (unsigned byte = 0; byte < 256; byte++)
        for (int k = 0; k < 4; k++) {
            n += 4;
            if (dotp_lut_a[byte * 4 + k] != gvo::lut_a(byte, k)) bad++;
            if (dotp_lut_b[byte * 4 + k] != gvo::lut_b(byte, k)) bad++;
            if (dotp_lut_ab[byte * 8 + k] != gvo::lut_a(byte, k)) bad++;
            if (dotp_lut_ab[byte * 8 + 4 + k] != gvo::lut_b(byte, k)) bad++;
        }
    for (unsigned nib = 0; nib < 16; nib++)
        for (int k = 0; k < 4; k++) {
            n += 1;
            if (na_lut[nib * 4 + k] != gvo::lut_na(nib, k)) bad++;
        }
    printf("lutcheck: %ld entries compared, %ld mismatches\n", n, bad);
    return bad ? 1 : 0;
}
