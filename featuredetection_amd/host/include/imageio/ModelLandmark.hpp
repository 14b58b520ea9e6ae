// ModelLandmark.hpp of the reference -- see imageio_all.hpp
#pragma once
#include "imageio/imageio_all.hpp"
