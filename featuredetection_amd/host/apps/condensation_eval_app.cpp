// condensation_eval_app -- the measurement step of the reference's particle-filter trackers (condensation::WvmSvmModel,
// as wired in faceTrackingApp): scores a list of samples on one frame.
//   usage: condensation_eval_app <config.cfg> <image.ppm|pgm> <samples.txt>
// config: the FaceFrontal-style `detectors.<name>` node of ffp_detect_app (firstClassifier pwvm, secondClassifier psvm,
// pyramid); samples.txt: one "x y size" per line.  Prints "<target 0|1> <weight>" per sample.
#include <cstdio>
#include <fstream>
#include <iostream>
#include "condensation/condensation_all.hpp"
#include "detection/detection_all.hpp"

using namespace imageprocessing;
using namespace classification;
using namespace condensation;
using boost::property_tree::ptree;
using std::make_shared;
using std::shared_ptr;
using std::string;

static cv::Mat read_pnm(const string& path) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f.is_open()) throw std::runtime_error("cannot open image " + path);
    string magic;
    int w, h, maxv;
    f >> magic >> w >> h >> maxv;
    f.get();
    if ((magic != "P5" && magic != "P6") || maxv != 255) throw std::runtime_error("only binary PGM/PPM with maxval 255 are supported");
    const int ch = magic == "P6" ? 3 : 1;
    cv::Mat img(h, w, CV_MAKETYPE(CV_8U, ch));
    f.read((char*)img.data, (size_t)w * h * ch);
    if (ch == 3)
        for (size_t i = 0; i < (size_t)w * h; ++i) std::swap(img.data[3 * i], img.data[3 * i + 2]);
    return img;
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s <config.cfg> <image.ppm|pgm> <samples.txt>\n", argv[0]);
        return 2;
    }
    try {
        ptree pt;
        boost::property_tree::read_info(string(argv[1]), pt);
        const ptree& node = pt.get_child("detectors").begin()->second;
        const ptree& imgpyr = node.get_child("pyramid");
        auto imgPyr = make_shared<ImagePyramid>((double)imgpyr.get<float>("incrementalScaleFactor", 0.9f), (double)imgpyr.get<float>("minScaleFactor", 0.09f),
                                                (double)imgpyr.get<float>("maxScaleFactor", 0.25f));
        imgPyr->addImageFilter(make_shared<GrayscaleFilter>());
        auto featureExtractor = make_shared<DirectPyramidFeatureExtractor>(imgPyr, imgpyr.get<int>("patch.width"), imgpyr.get<int>("patch.height"));
        featureExtractor->addPatchFilter(make_shared<HistEq64Filter>());
        auto wvm = ProbabilisticWvmClassifier::load(node.get_child("firstClassifier"));
        auto svm = ProbabilisticSvmClassifier::load(node.get_child("secondClassifier"));
        WvmSvmModel model(featureExtractor, wvm, svm);
        std::vector<shared_ptr<Sample>> samples;
        std::ifstream sf(argv[3]);
        int x, y, size;
        while (sf >> x >> y >> size) samples.push_back(make_shared<Sample>(x, y, size));
        auto image = make_shared<VersionedImage>(read_pnm(argv[2]));
        model.evaluate(image, samples);
        for (const auto& s : samples) std::printf("%d %.17g\n", s->isTarget() ? 1 : 0, s->getWeight());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
