// AggregatedFeaturesDetector.hpp of the reference -- see detection_all.hpp
#pragma once
#include "detection/detection_all.hpp"
