// RvmClassifier.hpp of the reference -- see classification_all.hpp
#pragma once
#include "classification/classification_all.hpp"
