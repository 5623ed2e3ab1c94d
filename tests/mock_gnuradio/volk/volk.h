/* TEST DOUBLE -- not VOLK.  lib/advanced_receiver_sb_cc_impl.cc includes <volk/volk.h> without calling anything from it; this empty
 * header only lets tests/test_boundary.py syntax-check that file unchanged. */
#ifndef MOCK_VOLK_H
#define MOCK_VOLK_H
#endif
