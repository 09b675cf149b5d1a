/* ganon_ibf_hash.h -- the constants of seqan3::interleaved_bloom_filter::hash_and_fit, ONE definition for every kernel, the host
 * and the tests (round 5 had five copies: gn_kernels.hip, gn_hibf.hip, gn_split.hip, gn_build.hip, host/verify.cpp).
 *
 * The reference calls the hash through SeqAn3 3.3.0 (GanonClassify.cpp:514/:553 -> counting_agent::bulk_count ->
 * membership_agent::bulk_contains -> hash_and_fit; GanonBuild.cpp:694 -> emplace); SeqAn3 is not in /root/reference, the numbers are
 * restated.  They are not arbitrary: each one follows from a closed form, and tests/test_oracle_kat.py derives all six
 * with 80-digit arithmetic and compares them with this header, with oracle/ganon_oracle.c and with what libganon_hip.so was built with:
 *     seed[0] = floor(2^64 / (e / 2))           seed[1] = floor(2^64 / sqrt 2) made odd (+1)      seed[2] = floor(2^64 / sqrt 3)
 *     seed[3] = floor(2^64 / (sqrt 5 / 2))      seed[4] = floor(2^63 / (3 pi / 5))                multiplier = floor(2^64 / golden ratio)
 *
 *     row(v, i) = mulhi64( (x ^ (x >> hash_shift)) * GN_IBF_MULTIPLIER, bin_size )   with x = v * seed[i], hash_shift = clz64(bin_size)
 */
#ifndef GANON_IBF_HASH_H
#define GANON_IBF_HASH_H

#define GN_IBF_MAX_HASH_FUNS 5
#define GN_IBF_SEED_LIST { 13572355802537770549ULL, 13043817825332782213ULL, 10650232656628343401ULL, 16499269484942379435ULL, 4893150838803335377ULL }
#define GN_IBF_MULTIPLIER 11400714819323198485ULL

#endif
