#include "../../cv_shim.hpp"
