// bs_maskgen.h — C++ declarations of the four entry points an application written against
// backscrub's library links to.  Signatures are those of /root/reference/lib/libbackscrub.h:13-39
// (plain `extern` C++ linkage, std::string / cv::Mat parameters), so app/deepseg.cc builds and
// links unchanged against libbsx.so + bs_maskgen_shim.o in place of libbackscrub.a.
// An application keeps including its own "libbackscrub.h"; this header exists so the shim and the
// tests in this repository do not need the reference tree.
#ifndef BS_MASKGEN_H_
#define BS_MASKGEN_H_

#include <cstddef>
#include <string>

#include <opencv2/core/core.hpp>
#include <opencv2/imgproc/imgproc.hpp>

// Version string of the inference engine behind the mask generator.
extern const char *bs_tensorflow_version(void);

// Create a mask generation context for frames of width x height (BGR, CV_8UC3).
// All four callbacks are optional; caller_ctx is handed back to them untouched.
// Returns nullptr after reporting through ondebug (or stderr) on any failure.
extern void *bs_maskgen_new(
	const std::string& modelname,
	size_t threads,
	size_t width,
	size_t height,
	void (*ondebug)(void *ctx, const char *msg),
	void (*onprep)(void *ctx),
	void (*oninfer)(void *ctx),
	void (*onmask)(void *ctx),
	void *caller_ctx
);

// Destroy a context (nullptr is ignored).
extern void bs_maskgen_delete(void *context);

// Turn one frame into a full-frame mask (255 = show background).  `mask` is re-pointed at a
// buffer owned by the context, valid until the next process/delete call on that context.
extern bool bs_maskgen_process(void *context, cv::Mat& frame, cv::Mat &mask);

#endif
