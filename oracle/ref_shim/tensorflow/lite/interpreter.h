#include "../../tflite_shim.hpp"
