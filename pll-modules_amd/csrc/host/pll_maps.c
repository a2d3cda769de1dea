/*
 * pll_maps.c -- character -> state-mask tables handed to pll_set_tip_states
 * (examples/spr-round/spr-round.c:193 uses pll_map_nt; src/util/models.c
 * selects a map per data type).  IUPAC nucleotide codes, the 20 amino acids in
 * the conventional A R N D C Q E G H I L K M F P S T W Y V order, binary.
 * Upper and lower case; gap / unknown = every state.
 */
#include "pll.h"

#define NT_ALL 15ULL
#define NT(c, C, m) [c] = (m), [C] = (m)

const pll_state_t pll_map_nt[256] = {
  NT('a', 'A', 1), NT('c', 'C', 2), NT('g', 'G', 4), NT('t', 'T', 8), NT('u', 'U', 8),
  NT('r', 'R', 5), NT('y', 'Y', 10), NT('s', 'S', 6), NT('w', 'W', 9),
  NT('k', 'K', 12), NT('m', 'M', 3), NT('b', 'B', 14), NT('d', 'D', 13),
  NT('h', 'H', 11), NT('v', 'V', 7), NT('n', 'N', NT_ALL), NT('o', 'O', NT_ALL),
  NT('x', 'X', NT_ALL), ['-'] = NT_ALL, ['?'] = NT_ALL
};

const pll_state_t pll_map_bin[256] = {
  ['0'] = 1, ['1'] = 2, ['-'] = 3, ['?'] = 3
};

#define AA_ALL ((1ULL << 20) - 1)
#define AA(c, C, bit) [c] = (1ULL << (bit)), [C] = (1ULL << (bit))

const pll_state_t pll_map_aa[256] = {
  AA('a', 'A', 0),  AA('r', 'R', 1),  AA('n', 'N', 2),  AA('d', 'D', 3),
  AA('c', 'C', 4),  AA('q', 'Q', 5),  AA('e', 'E', 6),  AA('g', 'G', 7),
  AA('h', 'H', 8),  AA('i', 'I', 9),  AA('l', 'L', 10), AA('k', 'K', 11),
  AA('m', 'M', 12), AA('f', 'F', 13), AA('p', 'P', 14), AA('s', 'S', 15),
  AA('t', 'T', 16), AA('w', 'W', 17), AA('y', 'Y', 18), AA('v', 'V', 19),
  ['b'] = (1ULL << 2) | (1ULL << 3),  ['B'] = (1ULL << 2) | (1ULL << 3),
  ['z'] = (1ULL << 5) | (1ULL << 6),  ['Z'] = (1ULL << 5) | (1ULL << 6),
  ['j'] = (1ULL << 9) | (1ULL << 10), ['J'] = (1ULL << 9) | (1ULL << 10),
  ['x'] = AA_ALL, ['X'] = AA_ALL, ['*'] = AA_ALL, ['-'] = AA_ALL, ['?'] = AA_ALL
};

/* Validity classes of the FASTA reader (test/src/tree/treemove-spr.c:178 and the other tree / binary test
 * programs pass it to pll_fasta_open; [libpll-2 knowledge]: 0 = illegal, reported with its line number;
 * 1 = legal sequence symbol; 2 = fatal; 3 = silently stripped white space).  The reader itself is
 * tier B3 (PLL_ERROR_NOT_IMPLEMENTED); the table exists so that those programs compile and link. */
#define FA_RANGE_26(first) [first] = 1, [first + 1] = 1, [first + 2] = 1, [first + 3] = 1, [first + 4] = 1, \
  [first + 5] = 1, [first + 6] = 1, [first + 7] = 1, [first + 8] = 1, [first + 9] = 1, [first + 10] = 1,   \
  [first + 11] = 1, [first + 12] = 1, [first + 13] = 1, [first + 14] = 1, [first + 15] = 1, [first + 16] = 1, \
  [first + 17] = 1, [first + 18] = 1, [first + 19] = 1, [first + 20] = 1, [first + 21] = 1, [first + 22] = 1, \
  [first + 23] = 1, [first + 24] = 1, [first + 25] = 1

const unsigned int pll_map_fasta[256] = {
  FA_RANGE_26('A'), FA_RANGE_26('a'),
  ['*'] = 1, ['-'] = 1, ['.'] = 1, ['?'] = 1, ['!'] = 1, ['0'] = 1, ['1'] = 1,
  ['\t'] = 3, ['\n'] = 3, ['\v'] = 3, ['\f'] = 3, ['\r'] = 3, [' '] = 3,
  [0] = 2
};
