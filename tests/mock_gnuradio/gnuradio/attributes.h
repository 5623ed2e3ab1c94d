/* TEST DOUBLE -- not GNU Radio; see io_signature.h in this directory. */
#ifndef MOCK_GNURADIO_ATTRIBUTES_H
#define MOCK_GNURADIO_ATTRIBUTES_H
#define __GR_ATTR_EXPORT __attribute__((visibility("default")))
#define __GR_ATTR_IMPORT __attribute__((visibility("default")))
#endif
