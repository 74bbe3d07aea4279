/* Summation-order choices of the pyramid arithmetic.
 *
 * The reference leaves this arithmetic to OpenCV (cv::resize, cv::sepFilter2D; not part of the
 * reference tree, version unpinned), so the order in which fp32 terms are added is a CHOICE of this
 * build, not a fact read from the reference.  Each choice is defined once, here, and
 * used by the HIP kernels (gslam_amd/csrc/) and by the CPU restatement the tests check them
 * against (test infrastructure, not part of the product) alike: a
 * fixture that settles an order changes this header and nothing else, and the bit-exact GPU-vs-
 * oracle tests keep holding.
 */
#ifndef GKLT_ORDER_H
#define GKLT_ORDER_H

/* Mean of the 2x2 block  a b / c d  (cv::resize to exactly half size, pyramid.cpp:60-66).
 * Default: the serial order of OpenCV's scalar code; -DGKLT_MEAN2X2_PAIRWISE: rows first, as a
 * SIMD path that adds the two rows' horizontal sums would. */
#ifdef GKLT_MEAN2X2_PAIRWISE
#define GKLT_MEAN2X2(a, b, c, d) ((((a) + (b)) + ((c) + (d))) * 0.25f)
#else
#define GKLT_MEAN2X2(a, b, c, d) (((((a) + (b)) + (c)) + (d)) * 0.25f)
#endif

#endif /* GKLT_ORDER_H */
