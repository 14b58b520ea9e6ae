// ffp_detect_app -- the plumbing of ffpDetectApp (ffpDetectApp.cpp:373-515 graph build, :548-620 loop)
// on top of the reference-shaped classes of this backend.  Reads a Boost-INFO style config with a
// `detectors` node exactly like ffpDetectApp/*.cfg (type fiveStageCascade | single), a binary PPM/PGM
// image, runs every face detector on the whole image and every feature detector inside the first face
// box, and prints one line per detection:  <detector> <landmark> x y w h probability
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unordered_map>
#include "detection/detection_all.hpp"

using namespace detection;
using namespace imageprocessing;
using namespace classification;
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
    if (ch == 3)  // PPM is RGB, the reference works on BGR
        for (size_t i = 0; i < (size_t)w * h; ++i) std::swap(img.data[3 * i], img.data[3 * i + 2]);
    return img;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <config.cfg> <image.ppm|pgm> [more images of a sequence ...]\n", argv[0]);
        return 2;
    }
    try {
        ptree pt;
        boost::property_tree::read_info(string(argv[1]), pt);
        std::vector<std::pair<string, shared_ptr<Detector>>> faceDetectors, featureDetectors;
        for (const auto& kv : pt.get_child("detectors")) {
            const ptree& node = kv.second;
            const string landmarkName = node.get<string>("landmark");
            const string type = node.get<string>("type");
            const ptree& imgpyr = node.get_child("pyramid");
            // float-typed like ffpDetectApp.cpp:407 (imgpyr.get<float>(...))
            auto imgPyr = make_shared<ImagePyramid>((double)imgpyr.get<float>("incrementalScaleFactor", 0.9f), (double)imgpyr.get<float>("minScaleFactor", 0.09f),
                                                    (double)imgpyr.get<float>("maxScaleFactor", 0.25f));
            imgPyr->addImageFilter(make_shared<GrayscaleFilter>());
            auto featureExtractor = make_shared<DirectPyramidFeatureExtractor>(imgPyr, imgpyr.get<int>("patch.width"), imgpyr.get<int>("patch.height"));
            shared_ptr<Detector> det;
            if (type == "fiveStageCascade") {
                auto firstClassifier = ProbabilisticWvmClassifier::load(node.get_child("firstClassifier"));
                auto secondClassifier = ProbabilisticSvmClassifier::load(node.get_child("secondClassifier"));
                const ptree& oeCfg = node.get_child("overlapElimination");
                auto oe = make_shared<OverlapElimination>(oeCfg.get<float>("dist", 5.0f), oeCfg.get<float>("ratio", 0.0f));
                featureExtractor->addPatchFilter(make_shared<HistEq64Filter>());
                auto swd = make_shared<SlidingWindowDetector>(firstClassifier, featureExtractor);
                det = make_shared<FiveStageSlidingWindowDetector>(swd, oe, secondClassifier);
            } else if (type == "single") {   // ffpDetectApp.cpp:427-500
                // one DirectPyramidFeatureExtractor per pyramid, one FilteringPyramidFeatureExtractor per classifier (:445)
                auto filteringExtractor = make_shared<FilteringPyramidFeatureExtractor>(featureExtractor);
                const string featurespace = node.get<string>("feature", "hq64");
                if (featurespace == "histeq") {
                    filteringExtractor->addPatchFilter(make_shared<HistogramEqualizationFilter>());
                } else if (featurespace == "whi") {
                    filteringExtractor->addPatchFilter(make_shared<WhiteningFilter>());
                    filteringExtractor->addPatchFilter(make_shared<HistogramEqualizationFilter>());
                    filteringExtractor->addPatchFilter(make_shared<ConversionFilter>(CV_32F, 1.0 / 127.5, -1.0));
                    filteringExtractor->addPatchFilter(make_shared<UnitNormFilter>(cv::NORM_L2));
                } else if (featurespace == "hq64") {
                    filteringExtractor->addPatchFilter(make_shared<HistEq64Filter>());
                } else if (featurespace != "gray") {
                    throw std::invalid_argument("unknown feature space " + featurespace);
                }
                if (node.count("patchFilter")) {   // ffpDetectApp.cpp:462-476
                    for (const auto& filterNode : node.get_child("patchFilter")) {
                        if (filterNode.first == "reshapingFilter") {
                            filteringExtractor->addPatchFilter(make_shared<ReshapingFilter>(filterNode.second.get_value<int>()));
                        } else if (filterNode.first == "conversionFilter") {
                            std::stringstream ss(filterNode.second.get_value<string>());
                            int type; double scaling;
                            ss >> type >> scaling;
                            filteringExtractor->addPatchFilter(make_shared<ConversionFilter>(type, scaling));
                        } else {
                            throw std::invalid_argument("unknown patch filter " + filterNode.first);
                        }
                    }
                }
                const ptree& classifierNode = node.get_child("classifier");
                const string classifierType = classifierNode.get_value<string>();
                shared_ptr<ProbabilisticClassifier> classifier;
                if (classifierType == "psvm") classifier = ProbabilisticSvmClassifier::load(classifierNode);
                else if (classifierType == "prvm") classifier = ProbabilisticRvmClassifier::load(classifierNode);
                else classifier = ProbabilisticWvmClassifier::load(classifierNode);   // "pwvm"
                det = make_shared<SlidingWindowDetector>(classifier, filteringExtractor);
            } else {
                throw std::invalid_argument("unknown detector type " + type);
            }
            det->landmark = landmarkName;
            (landmarkName == "face" ? faceDetectors : featureDetectors).emplace_back(kv.first, det);
        }
        if (argc > 3) {   // several images: the face detectors on all of them at once (detectFrames), one line per detection
            std::vector<cv::Mat> imgs;
            for (int a = 2; a < argc; ++a) imgs.push_back(read_pnm(argv[a]));
            for (auto& d : faceDetectors) {
                auto five = std::dynamic_pointer_cast<FiveStageSlidingWindowDetector>(d.second);
                if (!five) throw std::invalid_argument("several images need a fiveStageCascade face detector");
                const auto res = five->detectFrames(imgs);
                for (size_t f = 0; f < res.size(); ++f)
                    for (const auto& p : res[f]) {
                        cv::Rect b = p->getPatch()->getBounds();
                        std::printf("frame %zu %s %s %d %d %d %d %.17g\n", f, d.first.c_str(), d.second->landmark.c_str(), b.x, b.y, b.width, b.height, p->getProbability());
                    }
            }
            return 0;
        }
        cv::Mat img = read_pnm(argv[2]);
        std::vector<shared_ptr<ClassifiedPatch>> facePatches;
        for (auto& d : faceDetectors) {
            facePatches = d.second->detect(img);
            for (const auto& p : facePatches) {
                cv::Rect b = p->getPatch()->getBounds();
                std::printf("%s %s %d %d %d %d %.17g\n", d.first.c_str(), d.second->landmark.c_str(), b.x, b.y, b.width, b.height, p->getProbability());
            }
        }
        // the reference dereferences facePatches[0] unchecked (ffpDetectApp.cpp:591); here feature detectors are skipped without a face
        if (!facePatches.empty()) {
            cv::Rect faceBox = facePatches[0]->getPatch()->getBounds();
            for (auto& d : featureDetectors) {
                auto res = d.second->detect(img, faceBox);
                for (const auto& p : res) {
                    cv::Rect b = p->getPatch()->getBounds();
                    std::printf("%s %s %d %d %d %d %.17g\n", d.first.c_str(), d.second->landmark.c_str(), b.x, b.y, b.width, b.height, p->getProbability());
                }
            }
        }
    } catch (const std::invalid_argument& e) {
        std::fprintf(stderr, "invalid argument: %s\n", e.what());
        return 1;
    } catch (const std::runtime_error& e) {
        std::fprintf(stderr, "runtime error: %s\n", e.what());
        return 1;
    } catch (const std::logic_error& e) {
        std::fprintf(stderr, "logic error: %s\n", e.what());
        return 1;
    }
    return 0;
}
