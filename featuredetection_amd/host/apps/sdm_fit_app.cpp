// sdm_fit_app -- the two calls every SDM caller of the reference makes (detect-landmarks.cpp:272-276,
// sdmTracking.cpp:362,371): alignRigid + optimize, on the reference-shaped classes of this backend.
// usage: sdm_fit_app [--non-adaptive] <model.txt> <image.pgm> <x> <y> <w> <h> [landmarks.txt]   -> prints the 2L landmark coordinates;
// --non-adaptive: the `else` branches of SdmLandmarkModelFitting::optimize (SdmLandmarkModel.hpp:236-238,246-248) -- the descriptor
// parameters of the model file instead of the face-size adaptive ones (what the reference's shipped model needs);
// with a seventh argument the landmarks also go through imageio::SimpleModelLandmarkSink ("name x y" per line)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include "imageio/imageio_all.hpp"
#include "superviseddescent/superviseddescent_all.hpp"

using namespace superviseddescent;

int main(int argc, char** argv) {
    bool adaptive = true;
    if (argc > 1 && std::string(argv[1]) == "--non-adaptive") { adaptive = false; --argc; ++argv; }
    if (argc < 7) { std::fprintf(stderr, "usage: %s [--non-adaptive] model.txt image.pgm x y w h\n", argv[0]); return 2; }
    try {
        SdmLandmarkModel lmModel = SdmLandmarkModel::load(argv[1]);
        SdmLandmarkModelFitting modelFitter(lmModel, adaptive);
        std::ifstream f(argv[2], std::ios::binary);
        std::string magic; int w, h, maxv;
        f >> magic >> w >> h >> maxv; f.get();
        if (magic != "P5" || maxv != 255) throw std::runtime_error("need a binary PGM");
        cv::Mat imgGray(h, w, CV_8UC1);
        f.read((char*)imgGray.data, (size_t)w * h);
        cv::Rect faceBox(std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]));
        cv::Mat modelShape = lmModel.getMeanShape();
        modelShape = modelFitter.alignRigid(modelShape, faceBox);
        modelShape = modelFitter.optimize(modelShape, imgGray);
        for (int i = 0; i < modelShape.rows; ++i) std::printf("%.9g\n", modelShape.at<float>(i, 0));
        if (argc > 7) {   // like the landmark output of the reference's detect-landmarks app
            imageio::LandmarkCollection lms;
            const int L = modelShape.rows / 2;
            for (int i = 0; i < L; ++i)
                lms.insert(std::make_shared<imageio::ModelLandmark>(std::to_string(i), modelShape.at<float>(i, 0), modelShape.at<float>(i + L, 0)));
            imageio::SimpleModelLandmarkSink().add(lms, argv[7]);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
