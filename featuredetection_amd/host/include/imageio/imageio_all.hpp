// imageio/*.hpp of the reference, the part on the output side of the landmark path: Landmark,
// ModelLandmark, LandmarkCollection and the SimpleModelLandmarkSink ("name x y" per line, SimpleModelLandmarkSink.cpp:19-36).
// Header-only, no GPU involved.
#pragma once
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "fdcompat/cv.hpp"

namespace imageio {

// Landmark.hpp:27-120
class Landmark {
public:
    Landmark(const std::string& name, bool visible) : name(name), visible(visible) {}
    virtual ~Landmark() {}
    const std::string& getName() const { return name; }
    bool isVisible() const { return visible; }
    virtual float getX() const = 0;
    virtual float getY() const = 0;
private:
    std::string name;
    bool visible;
};

// ModelLandmark.hpp:20-95
class ModelLandmark : public Landmark {
public:
    explicit ModelLandmark(const std::string& name) : Landmark(name, false), x(0), y(0), z(0) {}
    ModelLandmark(const std::string& name, float x, float y, float z = 0) : Landmark(name, true), x(x), y(y), z(z) {}
    float getX() const override { return x; }
    float getY() const override { return y; }
    float getZ() const { return z; }
private:
    float x, y, z;
};

// LandmarkCollection.hpp:30-90
class LandmarkCollection {
public:
    void insert(std::shared_ptr<Landmark> landmark) { landmarks.push_back(landmark); }
    bool isEmpty() const { return landmarks.empty(); }
    bool hasLandmark(const std::string& name) const {
        for (const auto& l : landmarks) if (l->getName() == name) return true;
        return false;
    }
    const std::shared_ptr<Landmark> getLandmark(const std::string& name) const {
        for (const auto& l : landmarks) if (l->getName() == name) return l;
        throw std::invalid_argument("LandmarkCollection: there is no landmark with name '" + name + "'");
    }
    const std::vector<std::shared_ptr<Landmark>>& getLandmarks() const { return landmarks; }
private:
    std::vector<std::shared_ptr<Landmark>> landmarks;
};

// NamedLandmarkSink.hpp:24-37 (filename as std::string: Boost.Filesystem is absent)
class NamedLandmarkSink {
public:
    virtual ~NamedLandmarkSink() {}
    virtual void add(const LandmarkCollection& collection, std::string filename) = 0;
};

// SimpleModelLandmarkSink.cpp:19-36
class SimpleModelLandmarkSink : public NamedLandmarkSink {
public:
    void add(const LandmarkCollection& collection, std::string filename) override {
        std::ofstream outputFile(filename);
        if (!outputFile.is_open()) throw std::runtime_error("SimpleModelLandmarkSink: Error creating the output file " + filename);
        for (const auto& lm : collection.getLandmarks()) outputFile << lm->getName() << " " << lm->getX() << " " << lm->getY() << std::endl;
        outputFile.close();
    }
};

}  // namespace imageio
