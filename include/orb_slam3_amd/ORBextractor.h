/* ORBextractor.h — drop-in C++ facade with the reference's class name, namespace, constructor, call operator, getters
 * and public mvImagePyramid member (/root/reference/include/ORBextractor.h:43-109), forwarding to the HIP library through
 * the C ABI of include/orbx.h.  Header-only: put this directory before the reference's include/ on the include path and
 * link liborbx_hip.so; Frame.cc / Tracking.cc compile unchanged.
 *
 * Differences a caller can observe:
 *  - mvImagePyramid is refreshed after every call by copying the levels back from the GPU (each level is, like in the
 *    reference, a view into a buffer with a 19-px BORDER_REFLECT_101 frame).  SetExportPyramid(false) skips that copy when
 *    the stereo association also runs on the GPU (orbm_stereo_match reads the device-resident pyramid).
 *  - failures of the device (no GPU, out of memory) throw std::runtime_error; the reference cannot fail that way.
 */
#ifndef ORB_SLAM3_AMD_ORBEXTRACTOR_H
#define ORB_SLAM3_AMD_ORBEXTRACTOR_H
// This header REPLACES the reference's include/ORBextractor.h and takes its include guard, so that the `#include "ORBextractor.h"` lines inside the
// reference's own headers (KeyFrame.h:27, Tracking.h:34 - a quoted include looks in the including file's directory first, whatever the include path
// says) contribute nothing once this header has been seen.  If the reference's header was seen first the two classes would collide: fail with a reason.
#ifdef ORBEXTRACTOR_H
#error "the reference's include/ORBextractor.h was included before the drop-in ORBextractor.h: replace that file with this one, or force-include this header (-include); see INTEGRATION.md section 2"
#endif
#define ORBEXTRACTOR_H

#include <stdexcept>
#include <string>
#include <vector>
#include <list>
#include <opencv2/opencv.hpp>
#include "../orbx.h"

namespace ORB_SLAM3
{

class ORBextractor
{
public:
    enum {HARRIS_SCORE=0, FAST_SCORE=1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int deviceId = 0)
        : mpHandle(nullptr), mnLevels(nlevels), mfScaleFactor(scaleFactor), mbExportPyramid(true)
    {
        if (orbx_create(&mpHandle, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, deviceId) != ORBX_OK)
            throw std::runtime_error(std::string("ORBextractor (HIP): ") + orbx_last_error());
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        int umax[16];
        orbx_get_level_tables(mpHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                              mnFeaturesPerLevel.data(), umax);
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor() { orbx_destroy(mpHandle); }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image; mask is ignored (as in the reference).  Returns monoIndex, -1 if the image is empty.
    int operator()( cv::InputArray _image, cv::InputArray _mask,
                    std::vector<cv::KeyPoint>& _keypoints,
                    cv::OutputArray _descriptors, std::vector<int> &vLappingArea)
    {
        (void)_mask;
        if(_image.empty())
            return -1;
        cv::Mat image = _image.getMat();
        assert(image.type() == CV_8UC1 );
        const int cap = std::max(orbx_max_keypoints(mpHandle), 64);
        int n = 0, mono = -1;
        int rc = ORBX_E_CAPACITY;
        for (int attempt = 0; attempt < 2 && rc == ORBX_E_CAPACITY; attempt++) {      // the capacity is only known once the geometry is
            const int c = attempt == 0 ? cap : std::max(orbx_max_keypoints(mpHandle), n);
            mvKpBuf.resize(c); mvDescBuf.resize((size_t)c * 32);
            rc = orbx_extract(mpHandle, image.data, image.cols, image.rows, (int)image.step, vLappingArea[0], vLappingArea[1],
                              mvKpBuf.data(), mvDescBuf.data(), c, &n, &mono);
        }
        if (rc != ORBX_OK)
            throw std::runtime_error(std::string("ORBextractor (HIP): ") + orbx_last_error());
        if( n == 0 )
            _descriptors.release();
        else
            _descriptors.create(n, 32, CV_8U);
        _keypoints = std::vector<cv::KeyPoint>(n);
        cv::Mat descriptors = n ? _descriptors.getMat() : cv::Mat();
        for (int i = 0; i < n; i++) {
            const OrbxKeyPoint& k = mvKpBuf[i];
            cv::KeyPoint& o = _keypoints[i];
            o.pt.x = k.x; o.pt.y = k.y; o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = k.class_id;
            memcpy(descriptors.ptr(i), &mvDescBuf[(size_t)i * 32], 32);
        }
        if (mbExportPyramid) ExportPyramid();
        return mono;
    }

    int inline GetLevels(){ return mnLevels; }
    float inline GetScaleFactor(){ return mfScaleFactor; }
    std::vector<float> inline GetScaleFactors(){ return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors(){ return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares(){ return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares(){ return mvInvLevelSigma2; }

    std::vector<cv::Mat> mvImagePyramid;

    // ---- additions ----
    void SetExportPyramid(bool b) { mbExportPyramid = b; }
    void SetGaussianTaps(int variant) { orbx_set_gaussian_taps(mpHandle, variant); }   // 0: OpenCV >= 3.4/4.x, 1: OpenCV 3.2
    orbx_extractor* Handle() { return mpHandle; }    // for the orbm_* matchers (device-resident pyramid / descriptors)

protected:
    void ExportPyramid()
    {
        const int EDGE_THRESHOLD = 19;
        std::vector<uint8_t*> dst(mnLevels); std::vector<int> stride(mnLevels);
        std::vector<cv::Mat> framed(mnLevels);
        for (int level = 0; level < mnLevels; ++level) {
            int w = 0, h = 0;
            orbx_pyramid_level(mpHandle, 0, level, 0, nullptr, 0, &w, &h);           // sizes only (no copy)
            framed[level] = cv::Mat(cv::Size(w + EDGE_THRESHOLD*2, h + EDGE_THRESHOLD*2), CV_8UC1);
            mvImagePyramid[level] = framed[level](cv::Rect(EDGE_THRESHOLD, EDGE_THRESHOLD, w, h));
            dst[level] = mvImagePyramid[level].data; stride[level] = (int)mvImagePyramid[level].step;
        }
        // one device-to-host copy for the whole pyramid, then the reference's borders (src/ORBextractor.cc:1712-1736)
        if (orbx_pyramid_fetch(mpHandle, 0, 0, dst.data(), stride.data()) != ORBX_OK)
            throw std::runtime_error(std::string("ORBextractor (HIP): ") + orbx_last_error());
        for (int level = 0; level < mnLevels; ++level)
            cv::copyMakeBorder(mvImagePyramid[level], framed[level], EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD, EDGE_THRESHOLD,
                               cv::BORDER_REFLECT_101+cv::BORDER_ISOLATED);
    }

    orbx_extractor* mpHandle;
    int mnLevels;
    float mfScaleFactor;
    bool mbExportPyramid;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;
    std::vector<OrbxKeyPoint> mvKpBuf;
    std::vector<uint8_t> mvDescBuf;
};

} //namespace ORB_SLAM

#endif
