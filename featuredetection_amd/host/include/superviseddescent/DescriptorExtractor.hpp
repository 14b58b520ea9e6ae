// DescriptorExtractor.hpp of the reference -- see superviseddescent_all.hpp
#pragma once
#include "superviseddescent/superviseddescent_all.hpp"
