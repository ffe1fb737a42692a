// h5fileinfo_test.cpp -- include/hydroc_amd/h5fileinfo.h read the way the reference's own test reads a BEMIO file
// (tests/h5fileinfo_t01.cpp:20-24: H5FileInfo(h5fname, n).ReadH5Data(), GetRIRFTimeVector(), a copy of the HydroData), then every getter
// written to a flat file of doubles that tests/test_h5fileinfo.py compares with the committed fixtures.  No GPU involved.
//   h5fileinfo_test <file.h5> <num_bodies> <out.bin>      |      h5fileinfo_test --errors <file.h5>
#include <hydroc_amd/h5fileinfo.h>
#include <hydroc_amd/wave_types.h>

#include <cstdio>
#include <cstring>
#include <iostream>
#include <vector>

using namespace hydroc_amd;

int main(int argc, char* argv[]) {
    if (argc >= 3 && std::strcmp(argv[1], "--errors") == 0) {
        try {
            H5FileInfo("/nonexistent/dir/nothing.h5", 1).ReadH5Data();
            std::cout << "MISSING no-throw\n";
        } catch (const std::runtime_error& e) {
            std::cout << "MISSING " << e.what() << "\n";
        }
        try {
            H5FileInfo(argv[2], 7).ReadH5Data();  // more bodies than the file holds
            std::cout << "TOO_MANY no-throw\n";
        } catch (const std::runtime_error& e) {
            std::cout << "TOO_MANY " << e.what() << "\n";
        }
        HydroData d = H5FileInfo(argv[2], 1).ReadH5Data();
        try {
            d.GetRIRFVal(0, 6, 0, 0);
            std::cout << "RANGE no-throw\n";
        } catch (const std::out_of_range& e) {
            std::cout << "RANGE " << e.what() << "\n";
        }
        return 0;
    }
    if (argc >= 2 && std::strcmp(argv[1], "--spectrum") == 0) {
        // the spectrum helpers of wave_types.h (include/hydroc/wave_types.h:14-20): unsorted input is sorted in place
        std::vector<double> f = {0.31, 0.05, 0.125, 0.2, 0.08};
        const auto pm = PiersonMoskowitzSpectrumHz(f, 2.0, 8.0);
        const auto js = JONSWAPSpectrumHz(f, 2.0, 8.0);
        const auto jn = JONSWAPSpectrumHz(f, 2.0, 8.0, 2.0, true);
        for (size_t i = 0; i < f.size(); ++i) std::printf("%.17g %.17g %.17g %.17g\n", f[i], pm[i], js[i], jn[i]);
        return 0;
    }
    if (argc < 4) return 2;
    const std::string h5fname = argv[1];
    const int n               = std::atoi(argv[2]);

    HydroData infos = H5FileInfo(h5fname, n).ReadH5Data();
    auto rirf_time_vector = infos.GetRIRFTimeVector();
    HydroData infos2 = infos;  // (the reference's test copies the object)

    std::vector<double> out;
    auto put  = [&](double v) { out.push_back(v); };
    auto putv = [&](const std::vector<double>& v) { out.insert(out.end(), v.begin(), v.end()); };
    put(infos2.GetRhoVal());
    put(infos2.GetSimulationInfo().g);
    put(infos2.GetSimulationInfo().water_depth);
    for (int i = 0; i < 3; ++i) put(infos2.GetRIRFDims(i));
    putv(rirf_time_vector);
    for (int b = 0; b < n; ++b) {
        put(infos2.GetDispVolVal(b));
        putv(infos2.GetCGVector(b));
        putv(infos2.GetCBVector(b));
        const auto lin = infos2.GetLinMatrix(b);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) put(lin(i, j));
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) put(infos2.GetHydrostaticStiffnessVal(b, i, j));
        const auto am = infos2.GetInfAddedMassMatrix(b);
        put(am.rows());
        put(am.cols());
        for (int i = 0; i < am.rows(); ++i)
            for (int j = 0; j < am.cols(); ++j) put(am(i, j));
        for (int dof = 0; dof < infos2.GetRIRFDims(0); ++dof)
            for (int col = 0; col < infos2.GetRIRFDims(1); ++col)
                for (int s = 0; s < infos2.GetRIRFDims(2); ++s) put(infos2.GetRIRFVal(b, dof, col, s));
        const auto& body = infos2.GetBodyInfos()[static_cast<size_t>(b)];
        put(body.body_num);
        put(body.rirf_timestep);
        put(body.body_name == "body" + std::to_string(b + 1) ? 1.0 : 0.0);
        const auto& reg = infos2.GetRegularWaveInfos()[static_cast<size_t>(b)];
        put(static_cast<double>(reg.freq_list.size()));
        putv(reg.freq_list);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < reg.excitation_mag_matrix.cols(); ++j) put(reg.excitation_mag_matrix(i, j));
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < reg.excitation_phase_matrix.cols(); ++j) put(reg.excitation_phase_matrix(i, j));
        const auto& irr = infos2.GetIrregularWaveInfos()[static_cast<size_t>(b)];
        put(static_cast<double>(irr.excitation_irf_time.size()));
        putv(irr.excitation_irf_time);
        const auto ex = infos2.GetExcitationIRF(b);
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < ex.cols(); ++j) put(ex(i, j));
        put(infos2.GetExcitationIRFVal(b, 2, ex.cols() / 2) == ex(2, ex.cols() / 2) ? 1.0 : 0.0);
    }
    FILE* fp = std::fopen(argv[3], "wb");
    if (!fp) return 3;
    std::fwrite(out.data(), sizeof(double), out.size(), fp);
    std::fclose(fp);
    std::cout << "End" << std::endl;
    return 0;
}
