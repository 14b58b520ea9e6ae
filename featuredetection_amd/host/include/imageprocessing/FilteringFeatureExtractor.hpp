// FilteringFeatureExtractor.hpp of the reference -- see imageprocessing_all.hpp
#pragma once
#include "imageprocessing/imageprocessing_all.hpp"
