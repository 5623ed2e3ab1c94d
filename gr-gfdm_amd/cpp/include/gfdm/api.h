/* Symbol visibility for the GR-free kernel classes (role of include/gfdm/api.h in gr-gfdm). */
#ifndef INCLUDED_GFDM_API_H
#define INCLUDED_GFDM_API_H

#if defined(_WIN32)
#define GFDM_API
#else
#define GFDM_API __attribute__((visibility("default")))
#endif

#endif /* INCLUDED_GFDM_API_H */
