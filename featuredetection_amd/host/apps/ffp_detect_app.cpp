// ffp_detect_app -- the plumbing of ffpDetectApp (ffpDetectApp.cpp:373-515 graph build, :548-620 loop)
// on top of the reference-shaped classes of this backend.  Reads a Boost-INFO style config with a
// `detectors` node exactly like ffpDetectApp/*.cfg (type fiveStageCascade | single), a binary PPM/PGM
// image, runs every face detector on the whole image and every feature detector inside the first face
// box, and prints one line per detection:  <detector> <landmark> x y w h probability
//
// --gpus N (several images): image-shard data parallelism.  The process starts N copies of itself, one per GPU
// (FD_DEVICE = rank); rank r runs the face detectors on the images i with i mod N == r, all ranks exchange their detection records with
// ONE RCCL all-gather (fd_dist_gather_records) and rank 0 prints them in image order -- the same lines a single process prints.
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <algorithm>
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

// --gpus N: start one process per GPU and wait for them (the communicator id travels in a file).  Everything a single rank could
// fail on alone is checked HERE, before any rank joins the communicator -- a rank that dies while the others sit in
// ncclCommInitRank / ncclAllGather would leave them waiting forever: all images must be readable and N devices must exist
// (FD_DIST_ONE_DEVICE=1: every rank on device 0, for tests on a one-GPU box).  The first rank that exits with an error takes the
// others down, and so does a timeout (FD_DIST_TIMEOUT_S, default 600).
static int launch_ranks(int gpus, int argc, char** argv, const std::vector<char*>& args) {
    for (size_t a = 2; a < args.size(); ++a) {
        try { (void)read_pnm(args[a]); }
        catch (const std::exception& e) { std::fprintf(stderr, "runtime error: %s\n", e.what()); return 1; }
    }
    const bool oneDevice = std::getenv("FD_DIST_ONE_DEVICE") && std::atoi(std::getenv("FD_DIST_ONE_DEVICE")) != 0;
    {
        int ndev = 0;
        if (fd_device_count(&ndev) != FD_OK || ndev < 1) { std::fprintf(stderr, "runtime error: no usable GPU\n"); return 1; }
        if (!oneDevice && gpus > ndev) { std::fprintf(stderr, "invalid argument: --gpus %d, but %d device(s) are visible\n", gpus, ndev); return 1; }
    }
    uint8_t id[FD_DIST_ID_BYTES] = {0};
    if (gpus > 1 && fd_dist_unique_id(id) != FD_OK) { std::fprintf(stderr, "runtime error: librccl.so is not available\n"); return 1; }
    char path[] = "/tmp/fd_dist_id_XXXXXX";
    const int fdesc = mkstemp(path);
    if (fdesc < 0 || write(fdesc, id, sizeof(id)) != (ssize_t)sizeof(id)) { std::fprintf(stderr, "runtime error: cannot write %s\n", path); return 1; }
    close(fdesc);
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
        const pid_t pid = fork();
        if (pid == 0) {
            setenv("FD_DEVICE", oneDevice ? "0" : std::to_string(r).c_str(), 1);
            setenv("FD_DIST_RANK", std::to_string(r).c_str(), 1);
            setenv("FD_DIST_WORLD", std::to_string(gpus).c_str(), 1);
            setenv("FD_DIST_ID_FILE", path, 1);
            execv("/proc/self/exe", argv);
            std::perror("execv");
            _exit(127);
        }
        if (pid < 0) { std::perror("fork"); break; }
        kids.push_back(pid);
    }
    const char* te = std::getenv("FD_DIST_TIMEOUT_S");
    const double timeoutS = te && std::atof(te) > 0 ? std::atof(te) : 600.0;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = (int)kids.size() == gpus ? 0 : 1;
    size_t left = kids.size();
    bool killed = false;
    std::chrono::steady_clock::time_point tKill;
    while (left > 0) {
        int st = 0;
        const pid_t k = waitpid(-1, &st, WNOHANG);
        if (k > 0) {
            --left;
            kids.erase(std::remove(kids.begin(), kids.end(), k), kids.end());   // reaped: its pid may belong to someone else by now, never signal it again
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
        } else if (k < 0) {
            break;
        } else {
            usleep(2000);
        }
        const bool late = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutS;
        if ((rc != 0 || late) && !killed) {   // one rank failed (or nothing happens any more): the others would wait for it forever
            if (late) { std::fprintf(stderr, "runtime error: the ranks did not finish within %.0f s\n", timeoutS); rc = 1; }
            for (pid_t p : kids) kill(p, SIGTERM);
            killed = true;
            tKill = std::chrono::steady_clock::now();
        } else if (killed && std::chrono::duration<double>(std::chrono::steady_clock::now() - tKill).count() > 5.0) {
            // a rank blocked in a driver / RCCL call does not react to SIGTERM (the very hang this guards against): SIGKILL after a grace
            // period of 5 s, then at most 5 s more for the kernel to reap them -- the launcher never waits forever
            // (kids holds live children only: reaped ranks were removed above)
            for (pid_t p : kids) kill(p, SIGKILL);
            for (pid_t p : kids) {   // SIGKILL cannot be ignored: a blocking wait per remaining child ends, and leaves no zombie
                int st2 = 0;
                if (waitpid(p, &st2, 0) == p) --left;
            }
            kids.clear();
            break;
        }
    }
    unlink(path);
    (void)argc;
    return rc;
}

int main(int argc, char** argv) {
    int gpus = 0;
    string patchFile;   // --patches <file>: Detector::keepPatchData(true); every printed detection's patch (rows, cols, type as int32 + the pixels) is appended (one image)
    std::vector<char*> args;   // argv without --gpus N
    for (int a = 0; a < argc; ++a) {
        if (string(argv[a]) == "--gpus" && a + 1 < argc) { gpus = std::atoi(argv[++a]); continue; }
        if (string(argv[a]) == "--patches" && a + 1 < argc) { patchFile = argv[++a]; continue; }
        args.push_back(argv[a]);
    }
    if ((int)args.size() < 3 || (gpus != 0 && (gpus < 1 || gpus > 64 || args.size() < 4))) {
        std::fprintf(stderr, "usage: %s [--gpus N] [--patches <file>] <config.cfg> <image.ppm|pgm> [more images of a sequence ...]\n", argv[0]);
        return 2;
    }
    const char* rankEnv = std::getenv("FD_DIST_RANK");
    if (gpus > 0 && !rankEnv) return launch_ranks(gpus, argc, argv, args);
    const int rank = rankEnv ? std::atoi(rankEnv) : 0, world = rankEnv ? std::atoi(std::getenv("FD_DIST_WORLD")) : 1;
    argc = (int)args.size();
    argv = args.data();
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
        if (argc > 3 && rankEnv) {   // --gpus N: this rank's shard of the images, one gather of the detection records
            uint8_t id[FD_DIST_ID_BYTES] = {0};
            {
                std::ifstream f(std::getenv("FD_DIST_ID_FILE"), std::ios::binary);
                f.read((char*)id, sizeof(id));
                if (!f) throw std::runtime_error("cannot read the communicator id");
            }
            fd_dist* dist = nullptr;
            fdhost::check(fd_dist_init(fdhost::context(), rank, world, world > 1 ? id : nullptr, &dist));   // one rank: no communicator
            std::vector<cv::Mat> imgs;
            std::vector<int64_t> ids;
            const int nimg = argc - 2;
            for (int i = 0; i < nimg; ++i)
                if (fd_dist_owner(i, world) == rank) { imgs.push_back(read_pnm(argv[2 + i])); ids.push_back(i); }
            std::vector<fd_record> local;
            for (size_t di = 0; di < faceDetectors.size(); ++di) {
                auto five = std::dynamic_pointer_cast<FiveStageSlidingWindowDetector>(faceDetectors[di].second);
                if (!five) throw std::invalid_argument("several images need a fiveStageCascade face detector");
                const auto res = imgs.empty() ? std::vector<std::vector<shared_ptr<ClassifiedPatch>>>() : five->detectFrames(imgs);
                for (size_t f = 0; f < res.size(); ++f)
                    for (const auto& p : res[f]) {
                        const auto patch = p->getPatch();
                        fd_record r = {(double)ids[f], (double)di, (double)patch->getX(), (double)patch->getY(), (double)patch->getWidth(),
                                       (double)patch->getHeight(), 0.0, p->getProbability()};
                        local.push_back(r);
                    }
            }
            const int cap = 1 << 16;
            std::vector<fd_record> all((size_t)cap * world);
            int64_t nall = 0;
            int truncated = 0;
            fdhost::check(fd_dist_gather_records(dist, local.data(), (int)local.size(), cap, all.data(), (int64_t)all.size(), &nall, &truncated));
            fd_dist_destroy(dist);
            if (truncated) throw std::runtime_error("more than 65536 detections on one rank");
            if (rank == 0) {
                // the single-process order: detector by detector, frames ascending inside a detector
                std::stable_sort(all.begin(), all.begin() + nall, [](const fd_record& a, const fd_record& b) {
                    return a.detector != b.detector ? a.detector < b.detector : a.image < b.image;
                });
                for (int64_t i = 0; i < nall; ++i) {
                    const fd_record& r = all[(size_t)i];
                    const auto& d = faceDetectors[(size_t)r.detector];
                    const int w = (int)r.w, h = (int)r.h;
                    std::printf("frame %d %s %s %d %d %d %d %.17g\n", (int)r.image, d.first.c_str(), d.second->landmark.c_str(), (int)r.cx - w / 2,
                                (int)r.cy - h / 2, w, h, r.probability);
                }
            }
            return 0;
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
        std::FILE* pf = nullptr;
        if (!patchFile.empty()) {
            pf = std::fopen(patchFile.c_str(), "wb");
            if (!pf) throw std::invalid_argument("cannot write " + patchFile);
            for (auto& d : faceDetectors) d.second->keepPatchData(true);
            for (auto& d : featureDetectors) d.second->keepPatchData(true);
        }
        auto dumpPatch = [&](const ClassifiedPatch& p) {
            if (!pf) return;
            const cv::Mat& m = p.getPatch()->getData();
            const int32_t hdr[3] = {m.rows, m.cols, m.type()};
            std::fwrite(hdr, sizeof(hdr), 1, pf);
            for (int r = 0; r < m.rows; ++r) std::fwrite(m.ptr<unsigned char>(r), m.elemSize(), (size_t)m.cols, pf);
        };
        std::vector<shared_ptr<ClassifiedPatch>> facePatches;
        for (auto& d : faceDetectors) {
            facePatches = d.second->detect(img);
            for (const auto& p : facePatches) {
                cv::Rect b = p->getPatch()->getBounds();
                std::printf("%s %s %d %d %d %d %.17g\n", d.first.c_str(), d.second->landmark.c_str(), b.x, b.y, b.width, b.height, p->getProbability());
                dumpPatch(*p);
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
                    dumpPatch(*p);
                }
            }
        }
        if (pf) std::fclose(pf);
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
