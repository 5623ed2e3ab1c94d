/* TEST DOUBLE -- not GNU Radio; see ../io_signature.h. */
#ifndef MOCK_GNURADIO_DIGITAL_API_H
#define MOCK_GNURADIO_DIGITAL_API_H
#define DIGITAL_API
#endif
